#!/bin/bash
# Collect a round's profiling artifacts on a GPU box (run from the repository root through gpurun):
#   bash profiles/collect.sh r06
# GPU tests first, then the bench lines, the kernel-trace statistics (one context = isolated kernel times, default
# pipeline = under load) and the PMC counters -- kernel-trace statistics and counters in SEPARATE rocprofv3 runs, one
# counter group per pass, as MI355X_MICROARCH.md prescribes.  Outputs land in gpurun_out/<tag>/; profiles/summarize.py
# reduces them to the small files kept under profiles/<tag>/.
set -u
TAG=${1:-r06}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
(cd "$REPO" && timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > "$OUT/pytest_gpu.log"
cd /tmp; export TMPDIR=/tmp
B="python $REPO/bench.py"
SHORT="--steps 2 --warmup 1 --batches-per-step 8 --no-cpu-baseline --no-dense --no-power-probe"
# every kernel alone, per 64-buffer batch (the unit of DESIGN 3.2's table since round 1), then the bench's default command
# (128 buffers per launch, two contexts in flight): the correlation kernel's average duration there is what bench.py's
# roofline.kernel_ms must agree with
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_full_p1" -- $B $SHORT --batch 64 --pipeline 1 > "$OUT/stats_full_p1.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_full_default" -- $B $SHORT > "$OUT/stats_full_default.log" 2>&1
# the dense band (2-3 cells planted in every buffer) under the default command: where the per-cell chain's time goes
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_dense_default" -- $B $SHORT --dense-main > "$OUT/stats_dense_default.log" 2>&1
# complex<float> batches (k_xcorr_f16x3), every kernel alone
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c64_p1" -- $B $SHORT --batch 64 --pipeline 1 --input c64 > "$OUT/stats_c64_p1.log" 2>&1
# the tracker block (f4) and the streaming mode (configs[4]): their kernels' durations
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_track" -- $B --stage track --steps 20 --warmup 6 --no-cpu-baseline > "$OUT/stats_track.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_cut" -- python $REPO/tools/cut_probe.py > "$OUT/stats_cut.log" 2>&1      # lcs_track_cut: 64 cells x 980 symbols
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_stream" -- $B --stage stream --steps 100 --warmup 20 --no-cpu-baseline > "$OUT/stats_stream.log" 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc_$i" -- $B --steps 1 --warmup 0 --batch 64 --batches-per-step 4 --pipeline 1 --no-cpu-baseline --no-dense --no-power-probe > "$OUT/pmc_$i.log" 2>&1
done
# the fp16 kernel's HBM traffic (complex<float> batches): its own two passes
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmcf_$i" -- $B --steps 1 --warmup 0 --batch 64 --batches-per-step 4 --pipeline 1 --no-cpu-baseline --no-dense --no-power-probe --input c64 > "$OUT/pmcf_$i.log" 2>&1
done
# population parity (tools/parity_population.py: 1152 buffers, GPU chain against the oracle; round 6: + 512 buffers through fading multipath
# channels with the stage arrays of every decoded cell) and BASELINE configs[3]'s sweep on this library
(cd "$REPO" && timeout 1500 python tools/parity_population.py --out "$OUT/parity_population.json" > "$OUT/parity_population.log" 2>&1)
(cd "$REPO" && timeout 900 python tools/parity_population.py --groups channels --out "$OUT/parity_population_channels.json" > "$OUT/parity_population_channels.log" 2>&1)
(cd "$REPO" && timeout 900 python tools/parity_population.py --groups highband --out "$OUT/parity_population_highband.json" > "$OUT/parity_population_highband.log" 2>&1)
(cd "$REPO" && timeout 900 bash profiles/sweep_cli.sh "$TAG" > "$OUT/sweep_cli.log" 2>&1)
# the bench lines last: bench.py reports roofline.traffic only from a PMC summary taken from the running kernel sources
(cd "$REPO" && python profiles/summarize.py "$TAG" > /dev/null 2>&1)
timeout 300 $B --steps 20 --warmup 5 > "$OUT/bench_full_n1.json" 2> "$OUT/bench_full_n1.err"
# the multi-rank code path with ONE rank over RCCL (process group, device-identity all-gather, asynchronous record all-gather per
# step, timing collectives): its value must stay within 2 % of the plain line's
timeout 300 $B --gpus 1 --force-dist --dist-backend nccl --steps 20 --warmup 5 --no-cpu-baseline --no-dense > "$OUT/bench_forced_dist_n1.json" 2> "$OUT/bench_forced_dist.err"
timeout 200 $B --steps 20 --warmup 5 --input-host --no-cpu-baseline --no-dense > "$OUT/bench_full_n1_input_host.json" 2> "$OUT/bench_host.err"
timeout 120 $B --stage pss --no-cpu-baseline > "$OUT/bench_pss_n1.json" 2> "$OUT/bench_pss_n1.err"
timeout 120 $B --stage single --steps 200 --warmup 20 > "$OUT/bench_single_n1.json" 2> "$OUT/bench_single_n1.err"
timeout 120 $B --stage stream --steps 400 --warmup 20 > "$OUT/bench_stream_n1.json" 2> "$OUT/bench_stream_n1.err"
timeout 120 $B --stage track --steps 400 --warmup 40 > "$OUT/bench_track_n1.json" 2> "$OUT/bench_track_n1.err"
timeout 300 $B --steps 20 --warmup 3 --input c64 --no-cpu-baseline > "$OUT/bench_full_n1_c64_f16_kernel.json" 2> "$OUT/bench_c64.err"      # the driver's own command shape
# complex<float> batches that are dongle data, recognised on the device (lcs_set_float_batch_probe): the int8 route
timeout 300 $B --steps 20 --warmup 3 --input c64 --c64-probe --no-cpu-baseline > "$OUT/bench_full_n1_c64_probe_int8_route.json" 2> "$OUT/bench_c64p.err"
# the rate over a long run, and the tracker line (C++ loop: blocks alone, and every block from the dongle's bytes)
timeout 300 $B --steps 200 --warmup 5 --no-cpu-baseline > "$OUT/bench_full_n1_steps200.json" 2> "$OUT/bench_steps200.err"
# the CLI's band-7 grid: fc 2.6 GHz at the default 120 ppm -> n_f = 125 (24 template groups per buffer, 1.8 GB of xc_incoherent_single per batch)
timeout 400 $B --fc 2.6e9 --ppm 120 --steps 5 --warmup 1 > "$OUT/bench_full_n1_fc2600MHz_ppm120_nf125.json" 2> "$OUT/bench_nf125.err"
echo collected > "$OUT/done"
