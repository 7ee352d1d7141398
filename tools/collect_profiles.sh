# copy one final_run's outputs from gpurun_out/ into profiles/ under its tag:  tools/collect_profiles.sh r05_k
T=$1; G=gpurun_out; P=profiles
cp $G/${T}_bench.json $P/${T}_bench.json
cp $G/kernel_stats_${T}.txt $P/${T}_kernel_stats.txt
cp $(ls $G/step_timeline_${T}_*.txt | head -1) $P/${T}_step_timeline.txt
cp $G/hbm_traffic.txt $P/${T}_hbm_traffic.txt; cp $G/hbm_traffic.json $P/${T}_hbm_traffic.json
cp $G/mfma_util.txt $P/${T}_mfma_util.txt 2>/dev/null; cp $G/mfma_util.json $P/${T}_mfma_util.json 2>/dev/null
[ -f $G/replay_kernels_${T}.json ] && cp $G/replay_kernels_${T}.json $P/${T}_replay_kernels.json
for n in STFT_L41_enhance_graph front_L41_S3_N512_B128_graph front_DPCL_finetuning_graph front_DPCL_inference; do
  cp $G/cfg_${n}_stats.txt $P/${T}_cfg_${n}_kernel_stats.txt; cp $G/cfg_${n}_timeline.txt $P/${T}_cfg_${n}_timeline.txt
done
cp $G/cfg_STFT_L41_enhance_hbm_traffic.txt $P/${T}_cfg_STFT_L41_enhance_hbm_traffic.txt
cp $G/cfg_front_L41_S3_N512_B128_hbm_traffic.txt $P/${T}_cfg_front_L41_S3_N512_B128_hbm_traffic.txt
cp $G/${T}_other_configs.jsonl $P/${T}_other_configs.jsonl
cp $G/${T}_other_configs_reference_seeding.jsonl $P/${T}_other_configs_reference_seeding.jsonl 2>/dev/null
cp $G/${T}_gpu_suite.txt $P/${T}_gpu_suite.txt
ls $P | grep ${T}
