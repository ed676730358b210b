#!/bin/bash
# The first commands on a multi-GPU MI355X node (no such node has run this tree yet: one GPU per box so far), with what
# to expect from DESIGN.md section 6. Run from the repository root after `python -c "import __graft_entry__ as g; g.build()"`.
#   N=8 tools/first_8gpu_run.sh          (N: GPUs to use, default all)
set -u
N=${N:-$(python -c "import torch; print(torch.cuda.device_count())")}
export HSA_ENABLE_IPC_MODE_LEGACY=0 RTX_GATHER_TIMEOUT_MS=${RTX_GATHER_TIMEOUT_MS:-30000}
O=gpurun_out/first_${N}gpu; mkdir -p $O
echo "== 1. RCCL group tests + first contact (mismatch and silent-rank detection) on $N devices"
python -m pytest tests/test_gpu_multi.py -m gpu -q -k "rccl_group or first_contact or float_bands or contiguous" 2>&1 | tail -3 | tee $O/1_pytest.txt
echo "== 2. one process drives $N devices (rtx_create_multi): expect at 4K RGBA32F  N=2 ~0.33 ms (transfer-bound), 4 ~0.19, 8 ~0.16; rgba8 in also_measured"
python bench.py --gpus $N --also-bands > $O/2_bench_one_process.json 2> $O/2_bench_one_process.err; tail -c 1500 $O/2_bench_one_process.json; echo
echo "== 3. one process per GPU (rtx_create_rank; the driver's form): the same numbers within a few per cent; parity bit-identical"
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus $N > $O/3_bench_per_process.json 2> $O/3_bench_per_process.err; tail -c 1500 $O/3_bench_per_process.json; echo
echo "== 4. the main.cpp-style client on $N devices (RTX_DEVICES): FPS line + demo_frame.png"
make -C examples demo_main > /dev/null && RTX_DEVICES=$N examples/demo_main 120 | tee $O/4_demo.txt
echo "== read against DESIGN.md section 6: config.trace_ms_per_rank (a share ends with its torus tiles: ~157 us at N = 8 where 59 would be ideal),"
echo "   config.gather_GB_s_per_link (~150 GB/s per xGMI link expected), parity.vs_one_device_tracing_the_whole_frame == bit-identical"
