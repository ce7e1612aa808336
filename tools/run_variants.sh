#!/bin/bash
# tools/run_variants.sh <config> <kernel id> name [name ...]: prof_kernel.py against each tools/_bin/libmrt_<name>.so, two rounds
cfg=$1; kern=$2; shift; shift
for round in 1 2; do
	for n in "$@"; do
		echo -n "$n: "
		MRT_LIB_PATH=tools/_bin/libmrt_$n.so timeout -k 10 120 python3 tools/prof_kernel.py --config $cfg --kernel $kern --iters 12 || exit 1
	done
done
