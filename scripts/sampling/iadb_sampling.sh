#!/bin/bash
# Sampling command lines of the reference (scripts/sampling/{cat,celeba,church}_res{64,128}_test.sh),
# driven through the MI355X drop-in entry points.  Usage:
#   scripts/sampling/iadb_sampling.sh <dataset> <res> [ngpus]
# e.g. scripts/sampling/iadb_sampling.sh cat_res64 64 8
set -e
DATASET=${1:-cat_res64}; RES=${2:-64}; NGPU=${3:-1}
if [ "$RES" = "128" ]; then BS=200; TAU=0.2; else BS=500; TAU=1000; fi
RUN="python"
if [ "$NGPU" -gt 1 ]; then
  RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node $NGPU --master-addr 127.0.0.1 --master-port 29511"
fi
COMMON="--dataset=$DATASET --res=$RES --batch_size=$BS --train_or_test=test --nb_steps=250 --test_samples=30000"

# IADB with Gaussian white noise (paper figs 11/12, baseline)
$RUN iadb_bn.py $COMMON --noise_type=gaussian --scheduler_gamma=linear --scheduler_param=1 --out_channel=3 "${@:4}"
# ours: time-varying white -> blue noise
$RUN iadb_bn.py $COMMON --noise_type=gaussianBN --scheduler_gamma=sigmoid --scheduler_param=$TAU --out_channel=6 "${@:4}"
# DDIM baseline
$RUN ddim_diffusers.py --dataset_name="$DATASET" --train_or_test=test --eval_batch_size=$BS --test_samples=30000 \
  --resolution=$RES --random_flip --output_dir="ddim_$DATASET" --train_batch_size=2 --num_epochs=1000 \
  --gradient_accumulation_steps=1 --learning_rate=1e-4 --lr_warmup_steps=0 "${@:4}"
