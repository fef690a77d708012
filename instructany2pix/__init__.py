"""Import-name alias of the drop-in surface: `from instructany2pix import InstructAny2PixPipeline` (reference `instructany2pix/__init__.py:1`)
resolves to the MI355X-native pipeline. Nothing is implemented here: every name is forwarded, lazily, to `instructany2pix_amd` (importing this package touches no
GPU and no native code, like the package it forwards to)."""


def __getattr__(name):
    import instructany2pix_amd
    return getattr(instructany2pix_amd, name)
