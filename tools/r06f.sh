cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out instructany2pix_amd/plans
timeout 900 python bench.py --tune --save-plans gpurun_out/r06f_plans_all.txt --steps 20 --warmup 5 > gpurun_out/r06f_bench_tune.json 2> gpurun_out/r06f_bench_tune.err
cp bench_detail.json gpurun_out/r06f_bench_tune_detail.json
tail -1 gpurun_out/r06f_bench_tune.json | cut -c1-900
(echo "# kernel plan table of bench.py's five workloads (cfg 3 headline, cfg 2, cfg 5, the reference's 1024^2 defaults), measured in place on one MI355X by"; echo "# python bench.py --tune --save-plans (ia2p_autotune); format: M,N,K,conv,geglu,variant,splitk[,gn];"; cat gpurun_out/r06f_plans_all.txt) > instructany2pix_amd/plans/mi355x_bench.plans
IA2P_STAMP_LIB=instructany2pix_amd/libia2p_hip_stamp.so timeout 600 python tools/insitu_stamps.py 4 > gpurun_out/r06f_stamps_outproj.txt 2> gpurun_out/r06f_stamps_outproj.err
cat gpurun_out/r06f_stamps_outproj.txt; tail -3 gpurun_out/r06f_stamps_outproj.err
IA2P_STAMP_LIB=instructany2pix_amd/libia2p_hip_stamp.so timeout 600 python tools/insitu_stamps.py 2 > gpurun_out/r06f_stamps_ffout.txt 2> gpurun_out/r06f_stamps_ffout.err
cat gpurun_out/r06f_stamps_ffout.txt
bash tools/profile_round.sh r06f > gpurun_out/r06f_profile_round.log 2>&1
tail -5 gpurun_out/r06f_profile_round.log
head -12 gpurun_out/r06f_pmc_mfma.txt
