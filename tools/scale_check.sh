#!/bin/bash
# One-off scale check on the GPU box: a human-genome-size synthetic reference (3.1 Gb, 25 contigs, planted repeats), a thin read set, the
# command-line program against the compiled reference (oracle/_ref, mapping at -t1 from an index file it built itself at -t16).
# Usage: tools/scale_check.sh <genome_len> <contigs> <depth> <outdir>
set -u
GL=${1:-3100000000}; NC=${2:-25}; DEPTH=${3:-0.05}; OUT=${4:-gpurun_out/scale}
mkdir -p "$OUT"; W=$(mktemp -d /tmp/scale.XXXX)
t() { date +%s.%N; }
[ -x tools/gensim ] || gcc -O2 -o tools/gensim tools/gensim.c -lm
t0=$(t); tools/gensim genome 4001 "$GL" "$NC" 0.05 > "$W/ref.fa"; tools/gensim reads 4002 "$W/ref.fa" "$DEPTH" pacbio fa 20000 2000 > "$W/rd.fa"; t1=$(t)
echo "generate: $(awk "BEGIN{print $t1-$t0}") s; reads: $(grep -c '>' "$W/rd.fa")" | tee "$OUT/log.txt"
t0=$(t); MM_VERBOSE=1 minialign_amd/minialign -xpacbio "$W/ref.fa" "$W/rd.fa" > "$W/ours.sam" 2> "$OUT/ours.err"; echo "ours rc=$? $(awk "BEGIN{print $(t)-$t0}") s" | tee -a "$OUT/log.txt"
grep -E "main_align|M::main\]|index:" "$OUT/ours.err" | tee -a "$OUT/log.txt"
t0=$(t); minialign_amd/minialign -xpacbio -d "$W/ours.mai" "$W/ref.fa" 2>> "$OUT/ours.err"; minialign_amd/minialign -xpacbio "$W/ours.mai" "$W/rd.fa" 2>> "$OUT/ours.err" | grep -v '^@PG' | md5sum | tee -a "$OUT/log.txt"; echo "ours via .mai $(awk "BEGIN{print $(t)-$t0}") s" | tee -a "$OUT/log.txt"; rm -f "$W/ours.mai"
t0=$(t); oracle/_ref/minialign -xpacbio -t16 -d "$W/ref.mai" "$W/ref.fa" 2> "$OUT/ref.err"; echo "ref index rc=$? $(awk "BEGIN{print $(t)-$t0}") s" | tee -a "$OUT/log.txt"
t0=$(t); oracle/_ref/minialign -xpacbio -t1 "$W/ref.mai" "$W/rd.fa" > "$W/ref.sam" 2>> "$OUT/ref.err"; echo "ref map -t1 rc=$? $(awk "BEGIN{print $(t)-$t0}") s" | tee -a "$OUT/log.txt"
t0=$(t); oracle/_ref/minialign -xpacbio -t16 "$W/ref.mai" "$W/rd.fa" 2>> "$OUT/ref.err" | grep -vc '^@' | tee -a "$OUT/log.txt"; echo "ref map -t16 $(awk "BEGIN{print $(t)-$t0}") s" | tee -a "$OUT/log.txt"
grep -v '^@PG' "$W/ours.sam" | md5sum | tee -a "$OUT/log.txt"; grep -v '^@PG' "$W/ref.sam" | md5sum | tee -a "$OUT/log.txt"
echo "records: ours $(grep -vc '^@' "$W/ours.sam") ref $(grep -vc '^@' "$W/ref.sam"); mapped: ours $(grep -v '^@' "$W/ours.sam" | awk '$3!="*"' | wc -l)" | tee -a "$OUT/log.txt"
cmp <(grep -v '^@PG' "$W/ours.sam") <(grep -v '^@PG' "$W/ref.sam") | tee -a "$OUT/log.txt"
rm -rf "$W"
