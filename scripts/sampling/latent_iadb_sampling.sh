#!/bin/bash
# latent IADB sampling (reference: scripts/sampling/latent_iadb_{cat_res512,celeba_res256}_test.sh)
set -e
DATASET=${1:-cat_res512}; RES=${2:-512}
for NT in gaussian gaussianBN; do
  python latent_iadb_bn_diffusers.py --dataset_name="$DATASET" --resolution=$RES --train_or_test=test \
    --eval_batch_size=50 --test_samples=100 --random_flip --output_dir="latent_iadb_$DATASET" \
    --train_batch_size=256 --num_epochs=1000 --gradient_accumulation_steps=1 --learning_rate=1e-4 \
    --lr_warmup_steps=0 --out_channels=4 --noise_type=$NT "${@:3}"
done
