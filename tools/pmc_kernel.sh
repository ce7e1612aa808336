#!/bin/bash
# Counter passes over ONE kernel variant (tools/prof_kernel.py, or MRT_PMC_PROG=tools/prof_c4.py ...), through gpurun:
#   tools/pmc_kernel.sh <tag> --config C3 --kernel 11     ->  gpurun_out/<tag>/pmc_*/
# then, in the container:  python tools/pmc_table.py gpurun_out/<tag> [gpurun_out/<other tag> ...]
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
# MRT_PMC_PASSES="cycles insts" limits the passes (default: all)
pass() {
	name=$1; shift
	if [ -n "$MRT_PMC_PASSES" ] && [[ " $MRT_PMC_PASSES " != *" $name "* ]]; then return 0; fi
	rocprofv3 --pmc "$@" --kernel-trace -d "$out/pmc_$name" --output-format csv -- python3 $PROG "${ARGS[@]}" > "$out/pmc_$name.json" 2> "$out/pmc_$name.err"
	echo "pass $name done"
}
ARGS=("$@")
PROG=${MRT_PMC_PROG:-tools/prof_kernel.py}
python3 $PROG "$@" > "$out/plain.json"
cat "$out/plain.json"
pass cycles SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_WAVES
pass level SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
pass dcache SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_TC_STALL SQC_DCACHE_BUSY_CYCLES
pass icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass clock GRBM_GUI_ACTIVE
