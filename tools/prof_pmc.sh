#!/bin/bash
# usage: tools/prof_pmc.sh <outdir> <counters (space separated)> -- <python args...>
# Collect PMC counters for the reproject kernels in their own pass (no tracing flags besides kernel-trace).
out=$1; shift; ctrs=$1; shift; shift
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 "$@"
