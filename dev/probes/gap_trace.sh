#!/bin/bash
# where the GPU idles inside a Beyn pass: kernel trace of dev/probes/pass_only.py, gaps between consecutive dispatches of the second pass
set -e
TAG=${1:-r04}; shift || true
OUT=gpurun_out/gap_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python3 dev/probes/pass_only.py "$@" > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
python3 dev/probes/gap_analyse.py $OUT
