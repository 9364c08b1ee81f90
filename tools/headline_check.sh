#!/bin/bash
# Headline-shape run on the GPU box (BASELINE configs[3] shape on ONE GPU): a human-genome-size synthetic reference (3.1 Gb, 25 contigs, planted
# repeats) x PacBio-like reads at the given depth (3 = 9.2 Gb) through the command-line program, stage timings on stderr (MM_VERBOSE), and the head of
# the output compared with the compiled reference (oracle/_ref) at -t1 on the first reads; the reference at -t<all cores> on a larger sample for timing.
# Usage: tools/headline_check.sh <genome_len> <contigs> <depth> <outdir> [preset] [n_check_reads] [n_time_reads]
set -u
GL=${1:-3100000000}; NC=${2:-25}; DEPTH=${3:-3}; OUT=${4:-gpurun_out/headline}; PRE=${5:-pacbio}; NCHK=${6:-3000}; NTIME=${7:-40000}
KIND=pacbio; case "$PRE" in ont*) KIND=ont;; esac
mkdir -p "$OUT"; W=$(mktemp -d /tmp/headline.XXXX)
t() { date +%s.%N; }
[ -x tools/gensim ] || gcc -O2 -o tools/gensim tools/gensim.c -lm
t0=$(t); tools/gensim genome 0x5eed0001 "$GL" "$NC" 0.05 > "$W/ref.fa"; tools/gensim reads 0x5eed0002 "$W/ref.fa" "$DEPTH" $KIND fa 20000 2000 > "$W/rd.fa"; t1=$(t)
echo "generate: $(awk "BEGIN{print $t1-$t0}") s; reads: $(grep -c '>' "$W/rd.fa"); cores: $(nproc); mem: $(free -g | awk 'NR==2{print $2}') GB; tmp: $(df -h /tmp | awk 'NR==2{print $4}')" | tee "$OUT/log.txt"
t0=$(t); MM_VERBOSE=1 minialign_amd/minialign -x$PRE "$W/ref.fa" "$W/rd.fa" 2> "$OUT/ours.err" | tee >(head -c ${HEAD_BYTES:-600000000} > "$W/ours_head.sam"; cat > /dev/null) | wc -c > "$OUT/ours_bytes.txt"; echo "ours rc=${PIPESTATUS[0]} $(awk "BEGIN{print $(t)-$t0}") s, $(cat "$OUT/ours_bytes.txt") bytes" | tee -a "$OUT/log.txt"
grep -E "main_align|M::main\]|index:" "$OUT/ours.err" | tee -a "$OUT/log.txt"
awk -v n="$NCHK" '/^>/{c++} c<=n' "$W/rd.fa" > "$W/chk.fa"; awk -v n="$NTIME" '/^>/{c++} c<=n' "$W/rd.fa" > "$W/time.fa"
t0=$(t); oracle/_ref/minialign -x$PRE -t32 -d "$W/ref.mai" "$W/ref.fa" 2> "$OUT/ref.err"; echo "ref index -t32 rc=$? $(awk "BEGIN{print $(t)-$t0}") s" | tee -a "$OUT/log.txt"
t0=$(t); oracle/_ref/minialign -x$PRE -t1 "$W/ref.mai" "$W/chk.fa" > "$W/ref.sam" 2>> "$OUT/ref.err"; echo "ref map -t1 ($NCHK reads) rc=$? $(awk "BEGIN{print $(t)-$t0}") s" | tee -a "$OUT/log.txt"
NL=$(grep -vc '^@' "$W/ref.sam")
grep -v '^@' "$W/ref.sam" | md5sum | tee -a "$OUT/log.txt"; grep -v '^@' "$W/ours_head.sam" | head -n "$NL" | md5sum | tee -a "$OUT/log.txt"
cmp <(grep -v '^@' "$W/ref.sam") <(grep -v '^@' "$W/ours_head.sam" | head -n "$NL") | tee -a "$OUT/log.txt"
for T in 16 $(nproc); do
t0=$(t); oracle/_ref/minialign -x$PRE -t$T "$W/ref.mai" "$W/time.fa" 2> "$OUT/ref_t$T.err" | wc -c > /dev/null; echo "ref map -t$T ($NTIME reads, $(grep -v '>' "$W/time.fa" | wc -c) bases incl. newlines) $(awk "BEGIN{print $(t)-$t0}") s" | tee -a "$OUT/log.txt"
grep main_align "$OUT/ref_t$T.err" | tee -a "$OUT/log.txt"
done
rm -rf "$W"
