#!/bin/bash
# diagnostic: the last two parts of the ONT-like hg38-size set -- ours (several schedules) vs the compiled reference on the true two-part stream; the records of read $RD in full
OUT=${1:-gpurun_out/r4m}; W=/tmp/ontdiag; mkdir -p $OUT $W; RD=${RD:-r15.11589_}
G=tools/gensim; R=oracle/_ref/minialign
$G genome $((0x5eed0001)) 3100000000 25 0.05 > $W/ref.fa
for p in 14 15; do $G reads $((0x5eed0003)) $W/ref.fa 1.0 ont fa 20000 2000 $p 16 > $W/p$p.fa & done; wait
cat $W/p14.fa $W/p15.fa > $W/p1415.fa
( $R -xont.1dsq -t64 -d $W/ref.mai $W/ref.fa 2> $W/idx.err; $R -xont.1dsq -t1 $W/ref.mai $W/p1415.fa 2>/dev/null | grep -v '^@' | grep '^r15\.' > $W/ref15.sam ) &
minialign_amd/minialign -xont.1dsq -d $W/ours.mai $W/ref.fa 2> /dev/null
i=0
for env in "A=1" "A=2" "MM_K3_NO_JOBS=1 MM_K3_NO_RETRY_JOBS=1" "MM_LANES=1" "MM_ONE_SLAB_CLASS=1" "MM_K3_HOST_ROUNDS=1" "MM_BATCH_BASES=100000000"; do
	i=$((i+1))
	env $env timeout 300 minialign_amd/minialign -xont.1dsq $W/ours.mai $W/p1415.fa 2> $W/ours.$i.err | grep -v '^@' | grep '^r15\.' > $W/ours.$i.sam
	echo "run $i ($env): $(md5sum < $W/ours.$i.sam | cut -c1-12) $(grep -c . $W/ours.$i.sam) records; $(grep -c 're-run' $W/ours.$i.err)" | tee -a $OUT/runs.txt
	grep "^$RD" $W/ours.$i.sam | cut -f1-9 > $OUT/ours.$i.read.txt
done
wait
echo "reference: $(md5sum < $W/ref15.sam | cut -c1-12) $(grep -c . $W/ref15.sam) records" | tee -a $OUT/runs.txt
grep "^$RD" $W/ref15.sam | cut -f1-9 > $OUT/ref.read.txt
for j in 1 2 3 4 5 6 7; do cmp -s $W/ours.$j.sam $W/ref15.sam && echo "run $j == reference" || echo "run $j differs: $(diff <(cut -f1 $W/ours.$j.sam) <(cut -f1 $W/ref15.sam) | head -2 | tr '\n' ' ') lines: $(paste -d'|' <(md5sum < /dev/null) /dev/null >/dev/null; python3 -c "
a=open('$W/ours.$j.sam','rb').read().split(b'\n'); b=open('$W/ref15.sam','rb').read().split(b'\n')
print([x.split(b'\t')[0].decode() for x,y in zip(a,b) if x!=y][:5])")"; done | tee -a $OUT/runs.txt
