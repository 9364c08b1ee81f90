#!/bin/bash
# Soak on hard-repeat references (tools/gensim genomehard: repeat families, segmental duplications, satellite arrays) on the GPU box: seeded sets of varied size, contig count,
# read length and occurrence thresholds through the command-line program (several lanes, small batches: every batch boundary a carried-value hand-over) and the compiled
# reference (-t1); whole SAM compared.  Usage: tools/soak_hard.sh <outdir> [first_seed] [count]
OUT=${1:-gpurun_out/soak_hard}; S0=${2:-3000}; N=${3:-12}; mkdir -p "$OUT"; W=$(mktemp -d /tmp/soakh.XXXX); : > "$OUT/soak_hard.txt"
shapes=( "12000000 4 0.5 6000 2500 0.6 -xpacbio -f0.2,0.05,0.002" "30000000 25 0.45 20000 2000 0.5 -xpacbio" "8000000 1 0.6 10000 3000 1.0 -xpacbio -f0.1,0.02,0.001" "40000000 60 0.45 15000 5000 0.4 -xpacbio"
         "16000000 8 0.5 8000 3000 0.6 -xpacbio -k13 -w6" "20000000 12 0.45 20000 2000 0.5 -xpacbio -s200" "10000000 3 0.7 5000 2000 0.8 -xpacbio -f0.3,0.1,0.01" "24000000 200 0.4 12000 4000 0.4 -xpacbio.clr" )
envs=( "MM_BATCH_BASES=3000000 MM_LANES=3" "" "MM_BATCH_BASES=1000000 MM_LANES=4" "MM_DEVICE_CONTEXTS=2 MM_SLAB_GB=12 MM_BATCH_BASES=4000000" )
bad=0
for ((i=0; i<N; i++)); do
	set -- ${shapes[$(( i % ${#shapes[@]} ))]}; glen=$1; nc=$2; rep=$3; lm=$4; ls=$5; depth=$6; shift 6; opts="$*"; e=${envs[$(( i % ${#envs[@]} ))]}
	seed=$((S0 + 2 * i))
	tools/gensim genomehard $seed $glen $nc $rep > "$W/ref.fa"; tools/gensim reads $((seed + 1)) "$W/ref.fa" $depth pacbio fa $lm $ls > "$W/rd.fa"
	timeout 600 env $e minialign_amd/minialign $opts "$W/ref.fa" "$W/rd.fa" > "$W/o.sam" 2> "$W/o.err"; arc=$?
	timeout 900 oracle/_ref/minialign $opts -t1 "$W/ref.fa" "$W/rd.fa" > "$W/r.sam" 2> /dev/null; brc=$?
	a=$(grep -v '^@PG' "$W/o.sam" | md5sum | cut -c1-16); b=$(grep -v '^@PG' "$W/r.sam" | md5sum | cut -c1-16)
	st=ok; if [ "$a" != "$b" ] || [ "$arc" != 0 ] || [ "$brc" != 0 ]; then st=DIFF; bad=$((bad + 1)); tail -2 "$W/o.err" >> "$OUT/soak_hard.txt"; fi
	echo "$st seed=$seed genomehard=$glen/$nc/$rep reads=$lm/$ls x$depth ($(grep -c '^>' "$W/rd.fa") reads, $(grep -vc '^@' "$W/r.sam") records) opts='$opts' env='$e' ours=$a(rc=$arc) ref=$b(rc=$brc) $(grep 're-run' "$W/o.err" | sed 's/.*kernels/kernels/' | cut -c1-100)" | tee -a "$OUT/soak_hard.txt"
done
echo "mismatches: $bad of $N" | tee -a "$OUT/soak_hard.txt"; rm -rf "$W"
