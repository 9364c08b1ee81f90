#!/bin/bash
# Soak run on the GPU box: seeded synthetic sets of varied shape through the command-line program and the compiled reference (oracle/_ref, -t1);
# whole-SAM md5 compared.  Usage: tools/soak.sh <outdir> [first_seed] [count] [first_shape]
OUT=${1:-gpurun_out/soak}; S0=${2:-100}; N=${3:-12}; FIRST=${4:-0}          # FIRST: index of the first shape
mkdir -p "$OUT"; W=$(mktemp -d /tmp/soak.XXXX); : > "$OUT/soak.txt"
[ -x tools/gensim ] || gcc -O2 -o tools/gensim tools/gensim.c -lm
shapes=( "3000000 1 0.05 pacbio 20000 2000 30 -xpacbio" "2000000 8 0.30 pacbio 8000 3000 30 -xpacbio" "5000000 40 0.10 ont 0 0 20 -xont.1dsq" "1500000 3 0.50 pacbio 40000 10000 30 -xpacbio"
         "3000000 2 0.15 pacbio 3000 1500 40 -xpacbio.ccs" "2500000 5 0.20 ont 0 0 25 -xont.r9.4.1d" "4000000 1 0.02 pacbio 12000 6000 25 -xpacbio -k13 -w7" "2000000 6 0.25 pacbio 15000 4000 25 -xava"
         "3000000 4 0.10 pacbio 20000 2000 20 -xpacbio -c*" "1000000 2 0.70 pacbio 10000 3000 40 -xpacbio" "6000000 100 0.05 pacbio 25000 5000 15 -xpacbio -f0.1,0.02,0.002" "2000000 1 0.05 pacbio 60000 20000 30 -xont"
         "1000000 1 0.05 pacbio 400 200 20 -xpacbio -s20" "3000000 2 0.05 pacbio 100000 30000 20 -xpacbio" "500000 1 0.90 pacbio 8000 2000 30 -xpacbio" "4000000 2000 0.05 pacbio 6000 3000 15 -xpacbio"
         "2000000 3 0.20 pacbio 10000 3000 20 -xpacbio -k11 -w4" "2000000 3 0.20 pacbio 10000 3000 20 -xpacbio -s10 -m0.02" "2000000 2 0.10 pacbio 12000 3000 25 -xpacbio -Y20 -a1 -b1 -p1 -q1 -r0" "200000 5 0.10 pacbio 9000 2000 60 -xpacbio -c*"
         "2000000 4 0.30 ont 0 0 30 -xava -Opaf" "3000000 3 0.15 pacbio 15000 5000 20 -xpacbio -TSA,MD,NM,AS,XS" "2000000 2 0.40 pacbio 5000 2500 30 -xpacbio -Omaf" "1500000 1 0.10 pacbio 20000 8000 30 -xont.1dsq -W2000 -G1500" )
bad=0
for ((i=0; i<N; i++)); do
	set -- ${shapes[$(( (i + FIRST) % ${#shapes[@]} ))]}; glen=$1; nc=$2; rep=$3; prof=$4; lm=$5; ls=$6; depth=$7; shift 7; opts="$*"
	seed=$((S0 + 2 * i)); glen=$(( glen * ${SOAK_SCALE:-1} ))        # SOAK_SCALE: longer references (and, at equal depth, more reads)
	tools/gensim genome $seed $glen $nc $rep > "$W/ref.fa"
	fmt=fa; [ -n "$SOAK_FQ" ] && { fmt=fq; case "$opts" in *-O*) ;; *) opts="$opts -Q -TAS,NM,MD,XS,NH";; esac; }       # SOAK_FQ=1: FASTQ input, qualities kept, tags printed
	if [ "$prof" = ont ]; then tools/gensim reads $((seed + 1)) "$W/ref.fa" $depth ont $fmt > "$W/rd.fa"; else tools/gensim reads $((seed + 1)) "$W/ref.fa" $depth $prof $fmt $lm $ls > "$W/rd.fa"; fi
	timeout 300 env $SOAK_ENV minialign_amd/minialign $opts "$W/ref.fa" "$W/rd.fa" > "$W/o.sam" 2> "$W/o.err"; arc=$?     # SOAK_ENV: e.g. MM_BATCH_BASES=20000000 for many small batches on two lanes
	timeout 600 oracle/_ref/minialign $opts -t1 "$W/ref.fa" "$W/rd.fa" > "$W/r.sam" 2> /dev/null; brc=$?
	a=$(grep -v '^@PG' "$W/o.sam" | md5sum | cut -c1-16); b=$(grep -v '^@PG' "$W/r.sam" | md5sum | cut -c1-16)
	st=ok
	if [ "$brc" != 0 ]; then
		# the reference itself died (seen with -c: an extension that starts past the end of a sequence reads out of bounds): compare what it printed before
		n=$(( $(grep -vc '^@PG' "$W/r.sam") - 1 )); [ $n -lt 0 ] && n=0          # its buffer is cut wherever it was: the last line may be partial
		a2=$(grep -v '^@PG' "$W/o.sam" | head -n "$n" | md5sum | cut -c1-16); b=$(grep -v '^@PG' "$W/r.sam" | head -n "$n" | md5sum | cut -c1-16)
		if [ "$a2" = "$b" ] && [ "$arc" = 0 ]; then st="ok(reference died with rc=$brc after $n lines, identical up to there)"; else st=DIFF; bad=$((bad + 1)); fi
	elif [ "$a" != "$b" ] || [ "$arc" != 0 ]; then st=DIFF; bad=$((bad + 1)); tail -3 "$W/o.err" >> "$OUT/soak.txt"; fi
	echo "$st seed=$seed genome=$glen/$nc/$rep reads=$prof/$lm/$ls x$depth ($(grep -c '^[>@]r' "$W/rd.fa") reads) opts='$opts' ours=$a(rc=$arc) ref=$b $(grep 're-run' "$W/o.err" | sed 's/.*kernels/kernels/' | cut -c1-90)" | tee -a "$OUT/soak.txt"
done
echo "mismatches: $bad of $N" | tee -a "$OUT/soak.txt"
rm -rf "$W"
