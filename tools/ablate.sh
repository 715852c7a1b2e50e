#!/bin/bash
# Development tool: rebuild the library with one set of ablation macros at a time (results are
# WRONG by construction) and time the headline workload -- shows what each resource costs the
# kernel.  usage (on the GPU box): [ABLATIONS="A;B C;..."] tools/ablate.sh [bench args]
set -e
cd "$(dirname "$0")/.."
IFS=';' read -ra VARIANTS <<< "${ABLATIONS:-NONE;NOSTORE;NOBAR;NOLDS;NOFFT;NOFFT NOLDS;NOFFT NOLDS NOSTORE}"
for abl in "${VARIANTS[@]}"; do
    flags=""
    for a in $abl; do flags="$flags -DSPEC_ABL_$a"; done
    SPEC_EXTRA_HIPCC_FLAGS="$flags" python -m spectral_analyzer_amd.build --force > /dev/null 2>&1
    echo "== $abl: $(python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g lines/s  %.3f ms  frac %.3f  parity_ok=%s" % (d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["parity_spot_check"]["ok"]))')"
done
python -m spectral_analyzer_amd.build --force > /dev/null 2>&1
