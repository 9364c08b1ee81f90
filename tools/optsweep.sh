#!/bin/bash
W=$(mktemp -d /tmp/os.XXXX); tools/gensim genome 61 800000 3 0.2 > $W/ref.fa; tools/gensim reads 62 $W/ref.fa 6 pacbio fa 6000 2500 > $W/rd.fa
: > gpurun_out/optsweep.txt
while read -r o; do
	[ -z "$o" ] && continue
	timeout 200 minialign_amd/minialign -xpacbio $o $W/ref.fa $W/rd.fa > $W/o.sam 2> $W/oe; arc=$?
	timeout 300 oracle/_ref/minialign -xpacbio $o -t1 $W/ref.fa $W/rd.fa > $W/r.sam 2> /dev/null; brc=$?
	a=$(grep -v '^@PG' $W/o.sam | md5sum | cut -c1-12); b=$(grep -v '^@PG' $W/r.sam | md5sum | cut -c1-12)
	if [ $brc -gt 1 ]; then echo "refdied($brc) '$o' ours rc=$arc"; elif [ "$a" = "$b" ] && [ $(( arc != 0 )) = $(( brc != 0 )) ]; then echo "ok '$o' rc=$arc/$brc"; else echo "DIFF '$o' rc=$arc/$brc $a $b $(tail -1 $W/oe | cut -c1-80)"; fi
done <<'LIST' | tee -a gpurun_out/optsweep.txt | grep -v "^ok"
-k10
-k12 -w3
-k17
-k19 -w12
-k21 -w15
-k24
-k28 -w9
-k31 -w2
-w2
-w15
-w16
-w20
-a1
-a3
-a4 -b6
-a6 -b6
-b1
-b2
-b6
-p0
-p1
-p8
-p12
-q1
-q3 -r4,4
-q5 -r0
-r0
-r3,4
-r5,3
-r6,6
-Y11
-Y20
-Y80
-Y127
-s1
-s20
-s500
-s5000
-m0.01
-m0.1
-m0.9
-m0.99
-W200
-W1000
-W50000
-G200
-G1000
-G9000
-B2
-B8
-B20
-B28
-f0.5
-f0.001
-f0.9,0.5,0.1,0.05,0.01,0.005,0.001
-f0
-L1
-L7000
-eAA-1
-eAC2,CA2,GT1
-t4
-v
-v5
LIST
echo "sweep: $(grep -c "^ok" gpurun_out/optsweep.txt) ok of $(grep -c . gpurun_out/optsweep.txt)"
rm -rf $W
