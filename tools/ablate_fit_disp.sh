#!/bin/bash
# Instruction budget of fit_disp<4> at the C3 shape (50 000 x 500, ~batch + condition): the tuning build
# (make -C deseq2_amd/csrc ablate) with DSQ_FORCE_ITERS evaluations per gene and one component switched off at a time.
# bits: 1 per-sample log, 2 per-sample reciprocal, 4 lgamma/digamma over the distinct counts, 8 the p x p algebra
# (LU / inverse / trace), 16 the distinct-count sort; --nocr drops the Cox-Reid sums, closes, Gram build and algebra.
cd "$(dirname "$0")/.."
export DSQ_LIB=$PWD/deseq2_amd/libdeseq2_ablate.so DSQ_FORCE_ITERS=${ITERS:-12}
for abl in 0 1 2 3 4 8 16 31; do
  DSQ_ABLATE=$abl python tools/kbench.py --reps 3 2>/dev/null | grep KBENCH | sed "s/^/ablate=$abl /"
done
DSQ_ABLATE=0 python tools/kbench.py --reps 3 --nocr 2>/dev/null | grep KBENCH | sed "s/^/ablate=0 nocr /"
DSQ_ABLATE=7 python tools/kbench.py --reps 3 --nocr 2>/dev/null | grep KBENCH | sed "s/^/ablate=7 nocr /"
DSQ_ABLATE=23 python tools/kbench.py --reps 3 --nocr 2>/dev/null | grep KBENCH | sed "s/^/ablate=23 nocr /"
unset DSQ_FORCE_ITERS DSQ_ABLATE
DSQ_LIB=$PWD/deseq2_amd/libdeseq2_mi355x.so python tools/kbench.py --reps 3 2>/dev/null | grep KBENCH | sed "s/^/production /"
