#!/bin/bash
# The command-line program on a BASELINE shape, output to /dev/null, stage timings on stderr (MM_VERBOSE): the reference's own metric is
# `time minialign ... > out.sam` minus the index (README.md:42-53; stamps "loaded/built index" and "finished mapping", minialign.c:6417,6431).
# Usage: tools/cli_headline.sh <outdir> [genome_len contigs depth preset kind]      (defaults: the hg38-size x3 headline set)
set -u
OUT=${1:-gpurun_out/cli_headline}; GL=${2:-3100000000}; NC=${3:-25}; DEPTH=${4:-3}; PRE=${5:-pacbio}; KIND=${6:-pacbio}
mkdir -p "$OUT"; W=$(mktemp -d /tmp/clihead.XXXX)
[ -x tools/gensim ] || gcc -O2 -o tools/gensim tools/gensim.c -lm
tools/gensim genome 0x5eed0001 "$GL" "$NC" 0.05 > "$W/ref.fa"
for p in $(seq 0 15); do tools/gensim reads 0x5eed0002 "$W/ref.fa" "$DEPTH" $KIND fa 20000 2000 $p 16 > "$W/rd.$p" & done; wait
for p in $(seq 0 15); do cat "$W/rd.$p"; rm "$W/rd.$p"; done > "$W/rd.fa"
for rep in 1 2; do
  t0=$(date +%s.%N); env MM_VERBOSE=1 ${CLI_ENV:-} minialign_amd/minialign -x$PRE "$W/ref.fa" "$W/rd.fa" > /dev/null 2> "$OUT/run$rep.err"; echo "rc=$? wall $(awk "BEGIN{print $(date +%s.%N)-$t0}") s" | tee -a "$OUT/log.txt"
  grep -E "M::main_align|M::main\]" "$OUT/run$rep.err" | tee -a "$OUT/log.txt"
  awk '/loaded\/built index/{split($1,a,"::"); t0=a[3]+0} /finished mapping/{split($1,a,"::"); t1=a[3]+0} END{printf "map phase %.3f s\n", t1-t0}' "$OUT/run$rep.err" | tee -a "$OUT/log.txt"
done
if [ -n "${KEEP:-}" ]; then echo "$W"; else rm -rf "$W"; fi
