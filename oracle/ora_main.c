/* ora_main.c -- TEST INFRASTRUCTURE: command-line front end of the CPU oracle (`oracle/ora_minialign [-x<preset>] [options] ref.fa reads.fa`). */
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "ora_mm.h"
int main(int argc, char **argv)
{
	char const *files[16] = { 0 }; int nf = 0;
	char arg_line[4096] = "";
	for(int i = 0; i < argc; i++) { if(i) strcat(arg_line, " "); strncat(arg_line, argv[i], 1024); }
	om_opt_t o;
	if(om_opt_parse(&o, argc, (char const *const *)argv, files, 16, &nf) || nf < 2) { fprintf(stderr, "usage: ora_minialign [-x<preset>] [options] ref.fa reads.fa\n"); return 1; }
	o.arg_line = arg_line;
	double sec; uint64_t bases;
	int rc = om_main_files(&o, files, nf, stdout, &sec, &bases);
	fprintf(stderr, "[ora_minialign] mapped %lu bases in %.3f s (%.4f Gbases/s)\n", (unsigned long)bases, sec, bases / sec * 1e-9);
	return rc;
}
