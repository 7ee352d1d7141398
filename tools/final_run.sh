# Everything the round-end record needs, in one call:  T=r06_k bash tools/final_run.sh   -> gpurun_out/${T}_*
T=${T:-r06_k}
mkdir -p gpurun_out
python bench.py --quiet > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 400 gpurun_out/${T}_bench.json
TAG=_${T} bash tools/prof_step.sh
bash tools/pmc_traffic.sh
bash tools/pmc_mfma.sh
bash tools/prof_cfg.sh STFT_L41_enhance_graph pmc
bash tools/prof_cfg.sh front_L41_S3_N512_B128_graph pmc
bash tools/prof_cfg.sh front_DPCL_finetuning_graph
bash tools/prof_cfg.sh front_DPCL_inference
python tools/bench_configs.py > gpurun_out/${T}_other_configs.jsonl 2>/dev/null
# the CLI's default k-means seeding (--kmeans_seeding reference: the reference's host stream, bit-exact, host-bound) beside `fast`
python tools/bench_configs.py --seeding reference --only front_DPCL_inference,STFT_L41_enhance,STFT_L41_enhance_graph > gpurun_out/${T}_other_configs_reference_seeding.jsonl 2>/dev/null
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/${T}_gpu_suite.txt
cat gpurun_out/${T}_gpu_suite.txt
