#!/bin/bash
# CPU only: the plain-C oracle against the compiled reference on hard-repeat references (tools/gensim genomehard) of varied size, contig count and preset -- reads with
# dozens to hundreds of chains, rescue rounds, secondary records.  Usage: tools/oracle_soak_hard.sh <out.txt> [first_seed] [count]
OUT=${1:-/tmp/oracle_soak_hard.txt}; S0=${2:-5000}; N=${3:-12}
W=$(mktemp -d /tmp/osh.XXXX); : > "$OUT"
shapes=( "12000000 4 0.45 pacbio 8000 3000 0.04 -xpacbio" "20000000 40 0.50 pacbio 6000 2500 0.03 -xpacbio -f0.2,0.05,0.002" "8000000 2 0.60 ont 0 0 0.06 -xont.1dsq" "30000000 12 0.45 pacbio 15000 4000 0.02 -xpacbio"
         "10000000 6 0.45 pacbio 5000 2000 0.05 -xpacbio -TSA,MD,NM,XS" "16000000 100 0.40 pacbio 7000 2000 0.03 -xpacbio -Opaf" "9000000 3 0.70 pacbio 9000 3000 0.04 -xpacbio.ccs" "14000000 8 0.45 ont 0 0 0.03 -xont.r9.4.1d" )
bad=0
for ((i=0; i<N; i++)); do
	set -- ${shapes[$((i % ${#shapes[@]}))]}; glen=$1; nc=$2; rep=$3; prof=$4; lm=$5; ls=$6; depth=$7; shift 7; opts="$*"
	seed=$((S0 + 2 * i))
	tools/gensim genomehard $seed $glen $nc $rep > "$W/ref.fa"
	if [ "$prof" = ont ]; then tools/gensim reads $((seed + 1)) "$W/ref.fa" $depth ont fa > "$W/rd.fa"; else tools/gensim reads $((seed + 1)) "$W/ref.fa" $depth $prof fa $lm $ls > "$W/rd.fa"; fi
	timeout 1200 oracle/ora_minialign $opts "$W/ref.fa" "$W/rd.fa" > "$W/o.sam" 2> /dev/null; arc=$?
	timeout 600 oracle/_ref/minialign $opts -t1 "$W/ref.fa" "$W/rd.fa" > "$W/r.sam" 2> /dev/null; brc=$?
	a=$(grep -v '^@PG' "$W/o.sam" | md5sum | cut -c1-16); b=$(grep -v '^@PG' "$W/r.sam" | md5sum | cut -c1-16)
	if [ $brc -gt 1 ]; then st="refdied($brc)"; elif [ "$a" = "$b" ] && [ $(( arc != 0 )) = $(( brc != 0 )) ]; then st=ok; else st=DIFF; bad=$((bad + 1)); fi
	echo "$st seed=$seed genomehard=$glen/$nc/$rep reads=$prof/$lm/$ls x$depth ($(grep -c '^>' "$W/rd.fa") reads, $(grep -vc '^@' "$W/r.sam") records) opts='$opts' oracle=$a ref=$b" | tee -a "$OUT"
done
echo "oracle vs reference on hard repeats: mismatches $bad of $N" | tee -a "$OUT"
rm -rf "$W"
