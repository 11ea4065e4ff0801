#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r1        -> gpurun_out/r1_{stress,ref,prim,stream}/ (kernel trace + stats)
#                                       gpurun_out/r1_pmc_<COUNTER>/            (one counter per pass, bench at the stress shape)
#                                       gpurun_out/r1_pmcprim_<COUNTER>/        (north-star primitives)
# then, back in the container: python tools/summarize_profiles.py r1  (copies the summaries into profiles/).
# PMC passes are separate runs with --kernel-trace only, as the pool requires.
R=${1:-r6}
export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out
stats() { rocprofv3 --kernel-trace --stats -d $O/${R}_$1 -o runc --output-format csv -- "${@:2}" > $O/${R}_$1.log 2>&1; }
pmc()   { rocprofv3 --pmc $2 --kernel-trace -d $O/${R}_$1_$2 -o runc --output-format csv -- "${@:3}" > $O/${R}_$1_$2.log 2>&1; }
mkdir -p $O
stats stress python3 bench.py --steps 20 --warmup 3 --no-extras
stats ref    python3 bench.py --shape ref --steps 40 --warmup 3 --no-extras
stats prim   python3 tools/bench_primitives.py
stats stream python3 tools/profile_stream.py 1
stats crops  python3 tools/bench_crops.py
stats conv   python3 tools/bench_conv.py 1024 0
stats refiner python3 tools/refiner_loop.py 10
python3 tools/bench_fps.py 2>&1 | grep FPS > $O/${R}_fps.txt
# the own GEMM core against the vendor library, layer shape by layer shape (+ float64 check)
python3 tools/bench_linear_dma.py --check 2>&1 | grep -v amdgpu.ids > $O/${R}_gemm_ab.txt
# the split-bf16 switches on whole forwards, same job: fp32-MFMA kernels / split GEMMs / split GEMMs + split attention
{ python3 tools/ab_split.py 32; python3 tools/ab_split.py 32 12288 2048; } 2>&1 | grep -v amdgpu.ids > $O/${R}_split_ab.txt
# the sparse-conv stack layer by layer inside ordinary forwards (runner path, product library), and one launch from the inside
{ for a in "ref 32" "stress 32" "ref 6" "ref 1"; do python3 tools/conv_layers.py $a; done; } 2>&1 | grep -v amdgpu.ids > $O/${R}_conv_layers_runner.txt
{ for l in 0 1 2 3 4 5; do python3 tools/conv_stamps.py $l pair ref; done; } 2>&1 | grep -v "amdgpu.ids\|occupancy API" > $O/${R}_conv_stamps.txt
{ python3 tools/stream_b1.py 300 1; python3 tools/stream_b1.py 200 6; python3 tools/graph_timeline.py 1 6 32; } 2>&1 | grep -v amdgpu.ids > $O/${R}_small_calls.txt
python3 bench.py > $O/${R}_bench_default.log 2>&1; tail -1 $O/${R}_bench_default.log > $O/${R}_bench_default.jsonl
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  pmc pmc $c python3 bench.py --steps 4 --warmup 2 --no-extras
done
for c in FETCH_SIZE WRITE_SIZE; do
  pmc pmcprim $c python3 tools/bench_primitives.py
done
for c in SQ_VALU_MFMA_BUSY_CYCLES; do
  pmc pmcconv $c python3 tools/bench_conv.py 1024 0
done
for c in FETCH_SIZE WRITE_SIZE; do
  pmc pmccrops $c python3 tools/bench_crops.py
done
ls $O | grep "^${R}_" | tr '\n' ' '
