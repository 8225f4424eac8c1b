#!/bin/bash
# Evidence for the bf16-split GEMM (csrc/prn_gemm_split.hip): lab table against the fp32 MFMA kernel with fp64 error columns, sustained
# clock / power per kernel, matrix-pipe peaks, and the training / inference steps with the shader clock sampled.  -> gpurun_out/final/<tag>_*
TAG=${1:-r03_c}
cd "$(dirname "$0")/.."
O=gpurun_out/final; mkdir -p $O
L=tools/native/gemm_split_lab/split_lab.bin
{ for n in 6 9 3; do PRN_SPLIT_GEMM=0 SG_V=1 LAB_NPROD=$n LAB_REPS=200 timeout 300 $L; done; } > $O/${TAG}_split_gemm_lab.txt 2>&1
{ REPS=800 F=4096 bash tools/native/gemm_split_lab/clock_probe.sh; REPS=30000 F="s3 256->1024 @" bash tools/native/gemm_split_lab/clock_probe.sh; REPS=8000 F=fpn bash tools/native/gemm_split_lab/clock_probe.sh; } > $O/${TAG}_split_gemm_clock_probe.txt 2>&1
tools/native/mfma_peak/mfma_peak.bin 2 > $O/${TAG}_mfma_peak.txt 2>&1
{ echo "# training step (c3), PRN_SPLIT_MIN_TILES: 2500 = default policy, 300 = every launch the kernel wins alone"
  STEPS=40 bash tools/smi_ab.sh "PRN_SPLIT_GEMM=0" "PRN_SPLIT_GEMM=1" "PRN_SPLIT_GEMM=1 PRN_SPLIT_MIN_TILES=300" "PRN_SPLIT_GEMM=0" "PRN_SPLIT_GEMM=1" "PRN_SPLIT_GEMM=1 PRN_SPLIT_MIN_TILES=300"
  echo "# inference, 960-pixel workload (c5)"
  BENCH_ARGS="--workload c5" STEPS=40 bash tools/smi_ab.sh "PRN_SPLIT_GEMM=0" "PRN_SPLIT_GEMM=1" "PRN_SPLIT_GEMM=1 PRN_SPLIT_MIN_TILES=2500"
} > $O/${TAG}_split_gemm_step_ab.txt 2>&1
# the fp16 piece format (default): production path (prn_conv2d_fwd / prn_gemm_batched incl. the cutting kernels) against the fp32 kernel, warm and cold
{ for cfg in "PRN_SPLIT_GEMM=0" "PRN_SPLIT_GEMM=2 PRN_SPLIT_KIND=bf16" "PRN_SPLIT_GEMM=2 PRN_SPLIT_KIND=f16"; do
    echo "## $cfg, 300 back-to-back launches"; env $cfg LAB_ONLY=old LAB_REPS=300 timeout 300 $L | cut -c1-28,64-84 | grep -v "^shape\|NPROD"
    echo "## $cfg, 512 MB written between launches (operands from HBM, as inside a step)"; env $cfg LAB_ONLY=old LAB_COLD=1 timeout 300 $L | cut -c1-28,64-84 | grep -v "^shape\|NPROD"
  done; } > $O/${TAG}_split16_lab_production_path.txt 2>&1
python tools/acc16.py > $O/${TAG}_split16_accuracy_f16.txt 2>&1
PRN_SPLIT_KIND=bf16 python tools/acc16.py > $O/${TAG}_split16_accuracy_bf16.txt 2>&1
{ echo "# training step (c3): fp32 only / default (fp16 pieces, 300 tiles, kept images) / bf16 pieces / default without kept images / 2500 tiles"
  STEPS=60 bash tools/smi_ab.sh "PRN_SPLIT_GEMM=0" "DEFAULT=1" "PRN_SPLIT_KIND=bf16" "PRN_SPLIT_CACHE=0" "PRN_SPLIT_MIN_TILES=2500" "PRN_SPLIT_GEMM=0" "DEFAULT=1"
  for wl in c2 c5; do echo "# inference $wl"; BENCH_ARGS="--workload $wl" STEPS=40 bash tools/smi_ab.sh "PRN_SPLIT_GEMM=0" "DEFAULT=1" "PRN_SPLIT_KIND=bf16" "PRN_SPLIT_GEMM=0" "DEFAULT=1"; done
} > $O/${TAG}_split16_step_ab.txt 2>&1
tail -3 $O/${TAG}_split16_step_ab.txt
