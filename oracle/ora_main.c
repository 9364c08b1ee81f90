/* ora_main.c -- TEST INFRASTRUCTURE: command-line front end of the CPU oracle (`oracle/ora_minialign -x<preset> ref.fa reads.fa`). */
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "ora_mm.h"
int main(int argc, char **argv)
{
	char const *preset = "", *files[2] = { 0, 0 }; int nf = 0;
	char arg_line[4096] = ""; 
	for(int i = 0; i < argc; i++) { if(i) strcat(arg_line, " "); strncat(arg_line, argv[i], 1024); }
	for(int i = 1; i < argc; i++) { if(argv[i][0] == '-' && argv[i][1] == 'x') { preset = argv[i] + 2; } else if(argv[i][0] != '-' && nf < 2) { files[nf++] = argv[i]; } }
	if(nf < 2) { fprintf(stderr, "usage: ora_minialign -x<preset> ref.fa reads.fa\n"); return 1; }
	double sec; uint64_t bases;
	int rc = om_main(preset, files[0], files[1], stdout, arg_line, &sec, &bases);
	fprintf(stderr, "[ora_minialign] mapped %lu bases in %.3f s (%.4f Gbases/s)\n", (unsigned long)bases, sec, bases / sec * 1e-9);
	return rc;
}
