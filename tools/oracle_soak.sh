#!/bin/bash
# CPU only (build container): the plain-C oracle against the compiled reference on small seeded sets of varied shape and option lines -- the pinning of the
# oracle beyond the committed goldens.  Usage: tools/oracle_soak.sh <out.txt> [first_seed] [count]
OUT=${1:-/tmp/oracle_soak.txt}; S0=${2:-3000}; N=${3:-16}
W=$(mktemp -d /tmp/osk.XXXX); : > "$OUT"
shapes=( "300000 1 0.05 pacbio 6000 1500 2 -xpacbio" "200000 6 0.30 pacbio 3000 1200 3 -xpacbio" "400000 20 0.10 ont 0 0 1.5 -xont.1dsq" "150000 2 0.60 pacbio 8000 3000 3 -xpacbio"
         "300000 2 0.15 pacbio 1500 700 3 -xpacbio.ccs" "250000 3 0.20 ont 0 0 2 -xont.r9.4.1d" "300000 1 0.02 pacbio 5000 2500 2 -xpacbio -k13 -w7" "200000 4 0.25 pacbio 5000 1500 2 -xava -Opaf"
         "200000 3 0.10 pacbio 6000 1000 2 -xpacbio -c*" "100000 2 0.80 pacbio 4000 1000 3 -xpacbio -TSA,MD,NM" "400000 60 0.05 pacbio 7000 2000 1.5 -xpacbio -f0.1,0.02,0.002 -Omaf" "200000 1 0.05 pacbio 20000 8000 2 -xont -Q"
         "100000 1 0.05 pacbio 300 150 3 -xpacbio -s20" "150000 2 0.20 pacbio 4000 1500 2 -xpacbio -k11 -w4" "150000 2 0.20 pacbio 4000 1500 2 -xpacbio -s10 -m0.02 -P" "150000 2 0.10 pacbio 5000 1500 2 -xpacbio -Y20 -a1 -b1 -p1 -q1 -r0 -Oblast6" )
bad=0
for ((i=0; i<N; i++)); do
	set -- ${shapes[$((i % ${#shapes[@]}))]}; glen=$1; nc=$2; rep=$3; prof=$4; lm=$5; ls=$6; depth=$7; shift 7; opts="$*"
	seed=$((S0 + 2 * i))
	tools/gensim genome $seed $glen $nc $rep > "$W/ref.fa"
	if [ "$prof" = ont ]; then tools/gensim reads $((seed + 1)) "$W/ref.fa" $depth ont fq > "$W/rd.fq"; else tools/gensim reads $((seed + 1)) "$W/ref.fa" $depth $prof fq $lm $ls > "$W/rd.fq"; fi
	timeout 600 oracle/ora_minialign $opts "$W/ref.fa" "$W/rd.fq" > "$W/o.sam" 2> /dev/null; arc=$?
	timeout 300 oracle/_ref/minialign $opts -t1 "$W/ref.fa" "$W/rd.fq" > "$W/r.sam" 2> /dev/null; brc=$?
	a=$(grep -v '^@PG' "$W/o.sam" | md5sum | cut -c1-16); b=$(grep -v '^@PG' "$W/r.sam" | md5sum | cut -c1-16)
	if [ $brc -gt 1 ]; then st="refdied($brc)"; elif [ "$a" = "$b" ] && [ $(( arc != 0 )) = $(( brc != 0 )) ]; then st=ok; else st=DIFF; bad=$((bad + 1)); fi
	echo "$st seed=$seed genome=$glen/$nc/$rep reads=$prof/$lm/$ls x$depth ($(grep -c '^@r' "$W/rd.fq") reads) opts='$opts' oracle=$a ref=$b" | tee -a "$OUT"
done
echo "oracle vs reference: mismatches $bad of $N" | tee -a "$OUT"
rm -rf "$W"
