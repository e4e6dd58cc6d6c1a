/* kmahip_s2.c -- stage 2 of KMA as a pipe filter on an MI355X, in plain C99 over the C-ABI of libkmahip.so.
 *
 *     kma -i reads.fq -o x -t_db db -1t1 -s1 | kmahip_s2 -t_db db [-ex_mode]  >  s2.bin
 *
 * reads the S1 stream stage 1 of the reference writes (one record per read: seqlen, complen, N count, header length,
 * then the 2-bit words, the N positions and the header; savekmers.c:50-92 reads the same) and writes the S2 stream that
 * `kma ... -s2` writes (print_ankers, ankers.c:30-50; terminator kmers.c:257), byte for byte. It is the glue of
 * INTEGRATION.md section 2 as a program: everything device-side happens inside kmahip_scan_se.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kmahip.h"

static void die(const char *what) { fprintf(stderr, "kmahip_s2: %s: %s\n", what, kmahip_last_error()); exit(1); }

static void *xrealloc(void *p, size_t n) { p = realloc(p, n ? n : 1); if(!p) { fprintf(stderr, "kmahip_s2: out of memory\n"); exit(1); } return p; }

/* reverse complement of a 2-bit packed read (32 bases per word, first base in the top bits) and of its N list */
static void rc_packed(const uint64_t *seq, int len, const int32_t *N, int nN, uint64_t *rseq, int32_t *rN) {
	const int words = (len + 31) >> 5;
	memset(rseq, 0, (size_t) words * 8);
	for(int i = 0; i < len; ++i) {
		const int j = len - 1 - i;
		const uint64_t b = 3u - ((seq[j >> 5] >> (62 - ((j & 31) << 1))) & 3u);
		rseq[i >> 5] |= b << (62 - ((i & 31) << 1));
	}
	for(int i = 0; i < nN; ++i) rN[i] = len - 1 - N[nN - 1 - i];
}

int main(int argc, char **argv) {
	const char *prefix = NULL;
	int exhaustive = 0;
	for(int a = 1; a < argc; ++a) {
		if(!strcmp(argv[a], "-t_db") && a + 1 < argc) prefix = argv[++a];
		else if(!strcmp(argv[a], "-ex_mode")) exhaustive = 1;
		else { fprintf(stderr, "usage: kmahip_s2 -t_db <index prefix> [-ex_mode] < S1 stream > S2 stream\n"); return 2; }
	}
	if(!prefix) { fprintf(stderr, "kmahip_s2: -t_db is required\n"); return 2; }

	/* slurp the S1 stream into CSR arrays (words + one zero pad word per read, as kmahip_reads wants them) */
	int64_t n = 0, cap = 0, words = 0, wcap = 0, nN = 0, ncap = 0, hbytes = 0, hcap = 0;
	int max_len = 0;
	uint64_t *seq = NULL; int64_t *seq_off = NULL, *N_off = NULL, *h_off = NULL; int32_t *len = NULL, *Npos = NULL; char *hdr = NULL;
	int32_t head[4];
	while(fread(head, sizeof(int32_t), 4, stdin) == 4) {
		const int seqlen = head[0], complen = head[1], cnt = head[2], hl = abs(head[3]);
		if(head[3] < 0) { fprintf(stderr, "kmahip_s2: paired records in the stream: use the paired entry points (kmahip_scan_pe)\n"); return 1; }
		if(n + 2 > cap) { cap = cap ? 2 * cap : 1 << 16; seq_off = xrealloc(seq_off, (size_t) cap * 8); N_off = xrealloc(N_off, (size_t) cap * 8);
		                  h_off = xrealloc(h_off, (size_t) cap * 8); len = xrealloc(len, (size_t) cap * 4); }
		if(words + complen + 1 > wcap) { wcap = 2 * (words + complen + 1); seq = xrealloc(seq, (size_t) wcap * 8); }
		if(nN + cnt > ncap) { ncap = 2 * (nN + cnt) + 16; Npos = xrealloc(Npos, (size_t) ncap * 4); }
		if(hbytes + hl > hcap) { hcap = 2 * (hbytes + hl) + 64; hdr = xrealloc(hdr, (size_t) hcap); }
		seq_off[n] = words; N_off[n] = nN; h_off[n] = hbytes; len[n] = seqlen;
		if(fread(seq + words, 8, (size_t) complen, stdin) != (size_t) complen || fread(Npos + nN, 4, (size_t) cnt, stdin) != (size_t) cnt ||
		   fread(hdr + hbytes, 1, (size_t) hl, stdin) != (size_t) hl) { fprintf(stderr, "kmahip_s2: truncated S1 stream\n"); return 1; }
		words += complen; seq[words++] = 0; nN += cnt; hbytes += hl;
		if(seqlen > max_len) max_len = seqlen;
		++n;
	}
	if(!seq_off) { seq_off = xrealloc(NULL, 16); N_off = xrealloc(NULL, 16); h_off = xrealloc(NULL, 16); }
	seq_off[n] = words; N_off[n] = nN; h_off[n] = hbytes;

	kmahip_db *db; kmahip_ws *ws; kmahip_params par;
	if(kmahip_init(0) || kmahip_db_open(prefix, &db) || kmahip_ws_create(db, &ws)) die("open");
	kmahip_default_params(&par);
	par.exhaustive = exhaustive;

	kmahip_reads rd = { n, seq, seq_off, len, Npos, N_off, words, nN, max_len, NULL, NULL };
	int32_t *rc_flag = xrealloc(NULL, (size_t) (n + 1) * 4), *flag = xrealloc(NULL, (size_t) (n + 1) * 4), *T = NULL;
	int64_t *T_off = xrealloc(NULL, (size_t) (n + 1) * 8), T_cap = 8 * n + 1024;
	for(int tries = 0;; ++tries) {
		T = xrealloc(T, (size_t) T_cap * 4);
		kmahip_cands cd = { rc_flag, flag, T_off, T, T_cap };
		const int rc = kmahip_scan_se(db, ws, &rd, &par, &cd);
		if(rc == KMAHIP_OK) break;
		if(rc != KMAHIP_EOVERFLOW || tries > 6) die("kmahip_scan_se");
		if(T_off[n] > T_cap) T_cap = T_off[n] + 16;          /* KMAHIP_EOVERFLOW: T_off[n] holds the needed capacity */
	}

	/* S2 records in input order, exactly the bytes of print_ankers */
	uint64_t *rseq = xrealloc(NULL, (size_t) ((max_len + 31) / 32 + 2) * 8);
	int32_t *rN = xrealloc(NULL, (size_t) (max_len + 2) * 4);
	for(int64_t i = 0; i < n; ++i) {
		const int32_t nT = (int32_t) (T_off[i + 1] - T_off[i]);
		if(!nT) continue;
		const int complen = (int) (seq_off[i + 1] - seq_off[i] - 1), cnt = (int) (N_off[i + 1] - N_off[i]);
		const uint64_t *q = seq + seq_off[i];
		const int32_t *qN = Npos + N_off[i];
		if(flag[i] & 16) {                       /* the reverse-complemented read is the one passed on (savekmers.c:3049) */
			rc_packed(q, len[i], qN, cnt, rseq, rN);
			q = rseq; qN = rN;
		}
		const int32_t rec[7] = { len[i], complen, cnt, rc_flag[i], nT, (int32_t) (h_off[i + 1] - h_off[i]), flag[i] };
		fwrite(rec, sizeof(int32_t), 7, stdout);
		fwrite(q, 8, (size_t) complen, stdout);
		fwrite(qN, 4, (size_t) cnt, stdout);
		fwrite(T + T_off[i], 4, (size_t) nT, stdout);
		fwrite(hdr + h_off[i], 1, (size_t) (h_off[i + 1] - h_off[i]), stdout);
	}
	const int32_t end = (int32_t) -n;                /* kmers.c:257 */
	fwrite(&end, sizeof end, 1, stdout);
	fflush(stdout);
	kmahip_ws_destroy(ws);
	kmahip_db_close(db);
	return 0;
}
