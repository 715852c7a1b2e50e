#!/bin/bash
# Development tool: build EXPERIMENT libraries with one set of ablation macros each (results are WRONG by
# construction) and time a workload with every one of them -- shows what each resource costs the kernel.
# The variants are written beside the product library (lib/libspecgpu_abl<i>.so, selected through
# SPEC_LIB_VARIANT), never over it.
#   usage: [ABLATIONS="A;B C;..."] tools/ablate.sh build            (here: hipcc cross-compiles)
#          [ABLATIONS=...] tools/ablate.sh run [bench args]          (on the GPU box)
set -e
cd "$(dirname "$0")/.."
IFS=';' read -ra VARIANTS <<< "${ABLATIONS:-NONE;NOSTORE;NOBAR;NOLDS;NOFFT;NOFFT NOLDS;NOFFT NOLDS NOSTORE}"
mode=${1:-run}; shift || true
i=0
for abl in "${VARIANTS[@]}"; do
    flags=""
    for a in $abl; do flags="$flags -DSPEC_ABL_$a"; done
    if [ "$mode" = build ]; then
        python -m spectral_analyzer_amd.build --variant abl$i -- $flags > /dev/null
        echo "built abl$i: $abl"
    else
        echo "== $abl: $(SPEC_LIB_VARIANT=abl$i python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.4g lines/s  %.3f ms  frac %.3f  parity_ok=%s" % (d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["parity_spot_check"]["ok"]))')"
    fi
    i=$((i+1))
done
