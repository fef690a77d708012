"""MI355X-native denoise hot path of InstructAny2Pix (conditional SDXL UNet + DDIM loop).

Importing this package touches no GPU and no native code; the HIP library (libia2p_hip.so, built by
`python -m instructany2pix_amd.build`) is loaded on first use and there is no non-HIP fallback.
"""
from .config import UNetConfig, sdxl_base, sdxl_refiner, tiny

__all__ = ["InstructAny2PixPrior", "prior_config", "MODALITY", "HipGPT2Model", "DDPMScheduler", "HipCLIPTextModel", "SDXLTextEncoders", "UNetConfig", "sdxl_base", "sdxl_refiner", "tiny", "StableDiffusionXLImg2ImgPipeline", "EulerDiscreteScheduler", "InstructAny2PixPipeline", "HipUNet2DConditionModel", "DDIMScheduler",
           "SDXLDDIMPipeline", "StableDiffusionXLPipeline", "IPAdapterXL", "ImageProjModel", "HipAutoencoderKL", "EditRequest"]


def __getattr__(name):          # lazy: keep `import instructany2pix_amd` free of torch/ctypes work
    if name == "InstructAny2PixPipeline":
        from .pipeline import InstructAny2PixPipeline as v
    elif name == "HipUNet2DConditionModel":
        from .unet import HipUNet2DConditionModel as v
    elif name in ("InstructAny2PixPrior", "prior_config", "MODALITY", "HipGPT2Model"):
        from . import prior
        v = getattr(prior, name)
    elif name in ("DDIMScheduler", "EulerDiscreteScheduler", "DDPMScheduler"):
        from . import scheduler
        v = getattr(scheduler, name)
    elif name == "StableDiffusionXLImg2ImgPipeline":
        from .img2img import StableDiffusionXLImg2ImgPipeline as v
    elif name in ("SDXLDDIMPipeline", "StableDiffusionXLPipeline"):
        from . import ddim
        v = getattr(ddim, name)
    elif name in ("IPAdapterXL", "ImageProjModel"):
        from . import ip_adapter
        v = getattr(ip_adapter, name)
    elif name in ("HipCLIPTextModel", "SDXLTextEncoders"):
        from . import clip
        v = getattr(clip, name)
    elif name == "EditRequest":
        from .batch import EditRequest as v
    elif name == "HipAutoencoderKL":
        from .vae import HipAutoencoderKL as v
    else:
        raise AttributeError(name)
    return v
