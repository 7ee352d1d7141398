#!/bin/bash
# Every training entry point of the reference's command line (experiments/training/*.py), 8 synthetic batches each, in the order the
# recipes chain (pretraining -> front_<sep> -> enhance -> finetuning; STFT_<sep> -> enhance -> finetuning), eager and with --hip_graph:
# prints the last training loss and the best validation cost of every run; a replayed run must print what its eager twin prints, and no
# value may be nan.  (Found this way, round 6: a fine-tuning step that diverged only under --hip_graph at batch 8.)
#   bash tools/cli_smoke.sh            (needs a GPU; ~1 minute)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/adaptive-multispeaker-separation_amd
export AMS_LOG_DIR=$(mktemp -d /tmp/ams_cli_smoke.XXXXXX)
COMMON="--men --women --nb_speakers 2 --batch_size 8 --dataset synthetic --epochs 1 --synthetic_batches 8 --validation_step 5 --no_summaries"
bad=0
run() {
    name=$1; shift
    out=$(python -m experiments.training.$name "$@" 2>&1)
    if echo "$out" | grep -q "Traceback"; then echo "== $name: TRACEBACK"; echo "$out" | grep -v "^LOADED" | tail -4; bad=1; return; fi
    l=$(echo "$out" | grep "loss=" | tail -1 | sed 's/.*loss= *//; s/ .*//'); v=$(echo "$out" | grep "Best model with Validation" | tail -1 | awk '{print $NF}')
    printf "%-34s %-12s last loss=%-22s best validation=%s\n" "$name" "$(echo "$*" | grep -o -- '--hip_graph')" "$l" "$v"
    case "$l$v" in *nan*|*1e+100*) bad=1;; esac
}
last() { ls -dt $AMS_LOG_DIR/$1/* | head -1; }
run pretraining $COMMON --loss sdr+l2 --separation perfect --learning_rate 0.001 --filters 256 --max_pool 256 --beta 0.0 --regularization 0.0 --overlap_coef 1.0
P=$(last pretraining)
run pretraining $COMMON --with_max_pool --loss sdr+l2 --separation perfect --learning_rate 0.001 --filters 256 --max_pool 256 --beta 0.0 --regularization 0.0 --overlap_coef 1.0
FT="--learning_rate 0.0001 --optimizer RMSProp --nb_tries 1 --nb_steps 5 --beta_kmeans 10 --end_assign"
for g in "" "--hip_graph"; do
    run front_DPCL $COMMON --model_folder $P --learning_rate 0.001 $g
    run front_L41 $COMMON --model_folder $P --learning_rate 0.001 $g
done
D=$(last front_DPCL); Q=$(last front_L41)
for g in "" "--hip_graph"; do
    run front_DPCL_finetuning $COMMON --model_folder $D $FT --with_silence $g
    run front_L41_finetuning $COMMON --model_folder $Q $FT --with_silence $g
    run front_DPCL_enhance $COMMON --model_folder $D --learning_rate 0.001 --nb_tries 2 --nb_steps 3 --end_assign $g
    run front_L41_enhance $COMMON --model_folder $Q --learning_rate 0.001 --nb_tries 2 --nb_steps 3 --end_assign $g
done
E=$(last front_DPCL_enhance); EL=$(last front_L41_enhance)
for g in "" "--hip_graph"; do
    run front_DPCL_enhance_finetuning $COMMON --model_folder $E $FT --train prediction enhance $g
    run front_L41_enhance_finetuning $COMMON --model_folder $EL $FT --train prediction enhance $g
done
STFT="--window_size 512 --hop_size 256"
for g in "" "--hip_graph"; do
    run STFT_DPCL $COMMON $STFT --learning_rate 0.001 $g
    run STFT_L41 $COMMON $STFT --learning_rate 0.001 $g
done
SD=$(last STFT_DPCL); SL=$(last STFT_L41)
for g in "" "--hip_graph"; do
    run STFT_DPCL_enhance $COMMON $STFT --model_folder $SD --learning_rate 0.001 --nb_tries 2 --nb_steps 3 --end_assign $g
    run STFT_L41_enhance $COMMON $STFT --model_folder $SL --learning_rate 0.001 --nb_tries 2 --nb_steps 3 --end_assign $g
done
SE=$(last STFT_DPCL_enhance); SLE=$(last STFT_L41_enhance)
for g in "" "--hip_graph"; do
    run STFT_DPCL_finetuning $COMMON $STFT --model_folder $SE $FT --train prediction enhance $g
    run STFT_L41_finetuning $COMMON $STFT --model_folder $SLE $FT --train prediction enhance $g
done
rm -rf $AMS_LOG_DIR
[ $bad = 0 ] && echo "CLI smoke: all runs finite" || { echo "CLI smoke: FAILED"; exit 1; }
