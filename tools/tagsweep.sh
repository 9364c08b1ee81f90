#!/bin/bash
# output-side sweep: tags and formats one by one over a FASTQ read set with chimeric reads, product vs compiled reference (-t1)
W=$(mktemp -d /tmp/ts.XXXX); tools/gensim genome 71 600000 3 0.1 > $W/ref.fa; tools/gensim reads 72 $W/ref.fa 5 pacbio fq 5000 2000 > $W/rd.fq
# chimeras: glue pairs of reads
awk 'NR%4==1{n=$0} NR%4==2{s=$0} NR%4==0{q=$0; if(k%7==3 && ps!=""){print n"_chim some comment\n" ps s "\n+\n" pq q} else {print n"\n"s"\n+\n"q}; ps=s; pq=q; k++}' $W/rd.fq > $W/rc.fq
: > gpurun_out/tagsweep.txt
while read -r o; do
	[ -z "$o" ] && continue
	timeout 200 minialign_amd/minialign -xpacbio $o $W/ref.fa $W/rc.fq > $W/o.sam 2> $W/oe < /dev/null; arc=$?
	timeout 300 oracle/_ref/minialign -xpacbio $o -t1 $W/ref.fa $W/rc.fq > $W/r.sam 2> /dev/null < /dev/null; brc=$?
	a=$(grep -v '^@PG' $W/o.sam | md5sum | cut -c1-12); b=$(grep -v '^@PG' $W/r.sam | md5sum | cut -c1-12)
	if [ $brc -gt 1 ]; then echo "refdied($brc) '$o' ours rc=$arc"; elif [ "$a" = "$b" ] && [ $(( arc != 0 )) = $(( brc != 0 )) ]; then echo "ok '$o' rc=$arc/$brc $(grep -vc '^@' $W/o.sam) lines"; else echo "DIFF '$o' rc=$arc/$brc $a $b $(tail -1 $W/oe | cut -c1-80)"; fi
done <<'LIST' | tee -a gpurun_out/tagsweep.txt | grep -v "^ok"
-TNH
-TIH
-TAS
-TXS
-TNM
-TSA
-TMD
-TCO
-TRG -R@RG\tID:x\tSM:y
-R@RG\tID:onlyR
-TNH,IH,AS,XS,NM,SA,MD,CO
-TMD,SA -Q
-Q
-P
-P -TSA
-P -Q -TNM,MD
-A
-A -TXS
-Opaf
-Opaf -TAS
-Opaf -TID
-Opaf -TNM
-Opaf -TSQ
-Opaf -TCG
-Opaf -P -TAS,ID,NM,SQ,CG
-Oblast6
-Oblast6 -P
-Omaf
-Omaf -P
-Osam -TAS
-TZZ
-TAS;NM/MD:XS
LIST
echo "tag sweep: $(grep -c "^ok" gpurun_out/tagsweep.txt) ok of $(grep -c . gpurun_out/tagsweep.txt)"
rm -rf $W
