/*
 * minialign_cli.c -- the `minialign` command (host C): `minialign [-x preset] [opts] ref.fa reads.{fa,fq} > out.sam`
 * (reference: main, minialign.c:6451).  All work happens behind the C-ABI of include/minialign.h.
 */
#include "../../include/minialign.h"
int main(int argc, char **argv) { return mm_main(argc, argv); }
