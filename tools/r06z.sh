cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8) > gpurun_out/r06z_pytest.txt
tail -4 gpurun_out/r06z_pytest.txt
timeout 900 python bench.py --tune --save-plans gpurun_out/r06z_plans_all.txt --steps 20 --warmup 5 > gpurun_out/r06z_bench_tune.json 2> gpurun_out/r06z_bench_tune.err
cp bench_detail.json gpurun_out/r06z_bench_tune_detail.json
tail -1 gpurun_out/r06z_bench_tune.json | cut -c1-400
(echo "# kernel plan table of bench.py's five workloads (cfg 3 headline, cfg 2, cfg 5, the reference's 1024^2 defaults), measured in place on one MI355X by"; echo "# python bench.py --tune --save-plans (ia2p_autotune); format: M,N,K,conv,geglu,variant,splitk[,gn];"; cat gpurun_out/r06z_plans_all.txt) > instructany2pix_amd/plans/mi355x_bench.plans
bash tools/profile_round.sh r06z > gpurun_out/r06z_profile_round.log 2>&1
tail -3 gpurun_out/r06z_profile_round.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/r06z_bench_default_run.json 2> gpurun_out/r06z_bench_default_run.err
cp bench_detail.json gpurun_out/r06z_bench_default_run_detail.json
wc -c gpurun_out/r06z_bench_default_run.json
bash tools/ab_trees.sh "r05=.abtrees/r05 r06=. r06tune=.:--tune" "cfg3=;cfg2=--batch 1 --ctx 77;cfg5=--batch 8 --latent 96 --guidance 10;ref1024=--batch 1 --latent 128 --ctx 77" 2 > gpurun_out/r06z_same_box_r05_r06.txt 2>&1
cat gpurun_out/r06z_same_box_r05_r06.txt
