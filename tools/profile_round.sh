#!/bin/bash
# Collects the round's evidence for one bench.py workload on the GPU box (run through gpurun):
#   tools/profile_round.sh <tag> [bench.py args, e.g. --config C5]
#     ->  gpurun_out/<tag>/{bench.json, kt/, kt_bench.json, pmc_*/}
# then, back in the container:  python tools/summarize_prof.py gpurun_out/<tag> <tag> [--config C5]
# Counters are collected in separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; the guide's
# HBM section), each with --kernel-trace only.  The `clock` pass reads GRBM_GUI_ACTIVE: the effective shader
# clock of a dispatch is that count / 8 XCDs / its duration (guide, "DVFS give-back").
set -e
tag=${1:-round}
shift || true
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
python3 bench.py "$@" > "$out/bench.json" 2> "$out/bench.err"
tail -1 "$out/bench.json" | cut -c1-300
rocprofv3 --kernel-trace --stats -d "$out/kt" --output-format csv -- python3 bench.py --no-cpu-baseline "$@" > "$out/kt_bench.json" 2> "$out/kt.err"
pass() { # name, counters...
	name=$1; shift
	rocprofv3 --pmc "$@" --kernel-trace -d "$out/pmc_$name" --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "${BENCH_ARGS[@]}" > "$out/pmc_$name.json" 2> "$out/pmc_$name.err"
	echo "pass $name done"
}
BENCH_ARGS=("$@")
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass cycles SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
pass l2 TCC_HIT_sum TCC_MISS_sum
pass clock GRBM_GUI_ACTIVE
