/* kmahip_index.c -- `kma index -i templates.fsa [more.fsa ...] (or -batch list.txt) -o prefix [-k k]` on an MI355X: plain C99 over kmahip_index_build
 * (the k-mers are sorted and made unique on the device). Writes prefix.comp.b / .length.b / .seq.b / .name, which the reference
 * and libkmahip load alike.
 *
 *     kmahip_index -i genes.fsa -o genes [-k 16]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kmahip.h"

int main(int argc, char **argv) {
	const char *inputs[256], *out = NULL;
	int n_in = 0, k = 16;
	for(int a = 1; a < argc; ++a) {
		if(!strcmp(argv[a], "-i")) { while(a + 1 < argc && argv[a + 1][0] != '-' && n_in < 256) inputs[n_in++] = argv[++a]; }
		else if(!strcmp(argv[a], "-batch") && a + 1 < argc) {          /* index.c:351-401: a file that lists the inputs, one path a line */
			FILE *f = fopen(argv[++a], "rb");
			static char lines[256][4096];
			if(!f) { fprintf(stderr, "kmahip_index: cannot open %s\n", argv[a]); return 1; }
			while(n_in < 256 && fgets(lines[n_in], sizeof lines[0], f)) {
				size_t l = strlen(lines[n_in]);
				while(l && (lines[n_in][l - 1] == '\n' || lines[n_in][l - 1] == '\r' || lines[n_in][l - 1] == ' ' || lines[n_in][l - 1] == '\t')) lines[n_in][--l] = 0;
				if(l) { inputs[n_in] = lines[n_in]; ++n_in; }
			}
			fclose(f);
		}
		else if(!strcmp(argv[a], "-o") && a + 1 < argc) out = argv[++a];
		else if(!strcmp(argv[a], "-k") && a + 1 < argc) k = atoi(argv[++a]);
		else { fprintf(stderr, "usage: kmahip_index (-i <fasta> [<fasta> ...] | -batch <file of paths>) -o <index prefix> [-k <k-mer length, 4 ... 16>]\n"); return 2; }
	}
	if(!n_in || !out) { fprintf(stderr, "kmahip_index: -i and -o are required\n"); return 2; }
	if(kmahip_init(0) || kmahip_index_build(inputs, n_in, out, k)) { fprintf(stderr, "kmahip_index: %s\n", kmahip_last_error()); return 1; }
	return 0;
}
