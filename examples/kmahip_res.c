/* kmahip_res.c -- from the reads (or stage 1's stream) to the `.res` file on an MI355X, in plain C99 over the C-ABI of
 * libkmahip.so.
 *
 *     kmahip_res -t_db db -i reads.fq[.gz]  >  out.res                                  (stage 1 by kmahip_ingest_*)
 *     kma -i reads.fq -o x -t_db db -1t1 -s1 | kmahip_res -t_db db  >  out.res          (stage 1 by the reference)
 *
 * Single-end `-1t1` with KMA's defaults: stage 2 + 3a (kmahip_map_se), ConClave (kmahip_conclave_se), the row statistics
 * (kmahip_res_rows), the per-read traceback aligner (kmahip_align_trace), pile-up + consensus (kmahip_assemble) and the row
 * text (kmahip_res_line). The output equals the `.res` of `kma -i reads.fq -o out -t_db db -1t1 -t 1` byte for byte.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kmahip.h"

static void die(const char *what) { fprintf(stderr, "kmahip_res: %s: %s\n", what, kmahip_last_error()); exit(1); }
static void *xrealloc(void *p, size_t n) { p = realloc(p, n ? n : 1); if(!p) { fprintf(stderr, "kmahip_res: out of memory\n"); exit(1); } return p; }
static void *xcalloc(size_t n, size_t sz) { void *p = calloc(n ? n : 1, sz); if(!p) { fprintf(stderr, "kmahip_res: out of memory\n"); exit(1); } return p; }

int main(int argc, char **argv) {
	const char *prefix = NULL, *input = NULL;
	for(int a = 1; a < argc; ++a) {
		if(!strcmp(argv[a], "-t_db") && a + 1 < argc) prefix = argv[++a];
		else if(!strcmp(argv[a], "-i") && a + 1 < argc) input = argv[++a];
		else { fprintf(stderr, "usage: kmahip_res -t_db <index prefix> [-i reads.fq[.gz]]  (without -i: S1 stream on stdin)  > .res\n"); return 2; }
	}
	if(!prefix) { fprintf(stderr, "kmahip_res: -t_db is required\n"); return 2; }

	/* the S1 stream as CSR arrays (see kmahip_s2.c); headers are not needed for the `.res` */
	int64_t n = 0, cap = 0, words = 0, wcap = 0, nN = 0, ncap = 0;
	int max_len = 0;
	uint64_t *seq = NULL; int64_t *seq_off = NULL, *N_off = NULL; int32_t *len = NULL, *Npos = NULL;
	int32_t head[4];
	kmahip_ingest *ing = NULL;
	if(input) {
		/* stage 1 inside the library: parse, trim (KMA's defaults) and pack the whole file as one batch; the arrays stay
		 * owned by the reader, which is kept open until the end */
		kmahip_read_batch b;
		if(kmahip_ingest_open(input, NULL, NULL, &ing) || kmahip_ingest_next(ing, INT64_MAX, &b) || kmahip_ingest_status(ing)) die("ingest");
		n = b.reads.n_reads; words = b.reads.seq_words; nN = b.reads.N_total; max_len = b.reads.max_len;
		seq = (uint64_t *) b.reads.seq; seq_off = (int64_t *) b.reads.seq_off; N_off = (int64_t *) b.reads.N_off;
		len = (int32_t *) b.reads.len; Npos = (int32_t *) b.reads.N;
	}
	while(!input && fread(head, sizeof(int32_t), 4, stdin) == 4) {
		const int seqlen = head[0], complen = head[1], cnt = head[2], hl = abs(head[3]);
		if(head[3] < 0) { fprintf(stderr, "kmahip_res: paired records: not handled by this example\n"); return 1; }
		if(n + 2 > cap) { cap = cap ? 2 * cap : 1 << 16; seq_off = xrealloc(seq_off, (size_t) cap * 8); N_off = xrealloc(N_off, (size_t) cap * 8); len = xrealloc(len, (size_t) cap * 4); }
		if(words + complen + 1 > wcap) { wcap = 2 * (words + complen + 1); seq = xrealloc(seq, (size_t) wcap * 8); }
		if(nN + cnt > ncap) { ncap = 2 * (nN + cnt) + 16; Npos = xrealloc(Npos, (size_t) ncap * 4); }
		seq_off[n] = words; N_off[n] = nN; len[n] = seqlen;
		int left = hl, bad = fread(seq + words, 8, (size_t) complen, stdin) != (size_t) complen || fread(Npos + nN, 4, (size_t) cnt, stdin) != (size_t) cnt;
		char skip[4096];                                    /* the header is not needed here: read it away */
		while(!bad && left > 0) { const size_t got = fread(skip, 1, (size_t) (left < 4096 ? left : 4096), stdin); if(!got) bad = 1; left -= (int) got; }
		if(bad) { fprintf(stderr, "kmahip_res: truncated S1 stream\n"); return 1; }
		words += complen; seq[words++] = 0; nN += cnt;
		if(seqlen > max_len) max_len = seqlen;
		++n;
	}
	if(!input) {
		if(!seq_off) { seq_off = xrealloc(NULL, 16); N_off = xrealloc(NULL, 16); }
		seq_off[n] = words; N_off[n] = nN;
	}

	kmahip_db *db; kmahip_ws *ws; kmahip_params par; kmahip_db_info info;
	if(kmahip_init(0) || kmahip_db_open(prefix, &db) || kmahip_ws_create(db, &ws) || kmahip_db_get_info(db, &info)) die("open");
	kmahip_default_params(&par);
	const int64_t D = info.DB_size;

	/* stages 2 + 3a */
	kmahip_reads rd = { n, seq, seq_off, len, Npos, N_off, words, nN, max_len, NULL, NULL };
	int32_t *rc_flag = xcalloc((size_t) n + 1, 4), *flag = xcalloc((size_t) n + 1, 4);
	int64_t *T_off = xcalloc((size_t) n + 1, 8), T_cap = 8 * n + 1024;
	int32_t *T = NULL, *h_n = xcalloc((size_t) n + 1, 4), *h_best = xcalloc((size_t) n + 1, 4), *h_flag = xcalloc((size_t) n + 1, 4), *h_rc = xcalloc((size_t) n + 1, 4);
	int32_t *h_t = NULL, *h_sc = NULL, *h_s = NULL, *h_e = NULL;
	uint64_t *as = xcalloc((size_t) D, 8), *us = xcalloc((size_t) D, 8);
	kmahip_hits ht;
	for(int tries = 0;; ++tries) {
		T = xrealloc(T, (size_t) T_cap * 4); h_t = xrealloc(h_t, (size_t) T_cap * 4); h_sc = xrealloc(h_sc, (size_t) T_cap * 4);
		h_s = xrealloc(h_s, (size_t) T_cap * 4); h_e = xrealloc(h_e, (size_t) T_cap * 4);
		kmahip_cands cd = { rc_flag, flag, T_off, T, T_cap };
		kmahip_hits h = { h_n, h_best, h_flag, h_t, h_sc, h_s, h_e, as, us, h_rc };
		memset(as, 0, (size_t) D * 8); memset(us, 0, (size_t) D * 8);
		const int rc = kmahip_map_se(db, ws, &rd, &par, &cd, &h);
		ht = h;
		if(rc == KMAHIP_OK) break;
		if(rc != KMAHIP_EOVERFLOW || tries > 6) die("kmahip_map_se");
		if(T_off[n] > T_cap) T_cap = T_off[n] + 16;
	}
	kmahip_cands cd = { rc_flag, flag, T_off, T, T_cap };

	/* stage 3b */
	int32_t *pick = xcalloc((size_t) n + 1, 4), *pick_s = xcalloc((size_t) n + 1, 4), *pick_e = xcalloc((size_t) n + 1, 4);
	uint64_t *w_scores = xcalloc((size_t) D, 8);
	kmahip_conclave cc = { pick, pick_s, pick_e, w_scores, NULL, NULL, NULL };
	if(kmahip_conclave_se(db, ws, &rd, &cd, &ht, &cc)) die("kmahip_conclave_se");
	kmahip_res_row *rows = xcalloc((size_t) D, sizeof *rows);
	int64_t n_rows = 0;
	if(kmahip_res_rows(db, w_scores, 0.05, par.scoreT, rows, D, &n_rows)) die("kmahip_res_rows");
	uint8_t *ok = xcalloc((size_t) D, 1);
	for(int64_t r = 0; r < n_rows; ++r) ok[rows[r].template_id] = (uint8_t) rows[r].significant;

	/* stage 3c */
	int32_t *stats = xcalloc((size_t) n * 10 + 10, 4), *n_ops = xcalloc((size_t) n + 1, 4);
	int64_t *ops_off = xcalloc((size_t) n + 1, 8), ops_cap = 8 * n + 1024, need = 0;
	uint32_t *ops = NULL;
	kmahip_traces tr;
	for(int tries = 0;; ++tries) {
		ops = xrealloc(ops, (size_t) ops_cap * 4);
		kmahip_traces t = { stats, ops_off, n_ops, ops, ops_cap };
		const int rc = kmahip_align_trace(db, ws, &rd, h_rc, pick, ok, &par, &t, &need);
		tr = t;
		if(rc == KMAHIP_OK) break;
		if(rc != KMAHIP_EOVERFLOW || tries > 3) die("kmahip_align_trace");
		ops_cap = need + 16;
	}
	int64_t *cover = xcalloc((size_t) D, 8), *aln_len = xcalloc((size_t) D, 8), *depth = xcalloc((size_t) D, 8), *asm_len = xcalloc((size_t) D, 8);
	kmahip_assembly as3 = { cover, aln_len, depth, asm_len, NULL, NULL, 0, 0 };
	if(kmahip_assemble(db, ws, &rd, h_rc, pick, &tr, 0, 1, 0.05, &as3)) die("kmahip_assemble");

	/* the `.res` file: names come from <prefix>.name, one per line, in template order */
	char path[4096];
	snprintf(path, sizeof path, "%s.name", prefix);
	FILE *names = fopen(path, "r");
	if(!names) { fprintf(stderr, "kmahip_res: cannot open %s\n", path); return 1; }
	fputs("#Template\tScore\tExpected\tTemplate_length\tTemplate_Identity\tTemplate_Coverage\tQuery_Identity\tQuery_Coverage\tDepth\tq_value\tp_value\n", stdout);
	char *name = xrealloc(NULL, 1 << 16), *line = xrealloc(NULL, (1 << 16) + 512);
	int64_t r = 0;
	for(int64_t t = 1; t < D && fgets(name, 1 << 16, names); ++t) {
		name[strcspn(name, "\n")] = 0;
		while(r < n_rows && rows[r].template_id < t) ++r;
		if(r < n_rows && rows[r].template_id == t && rows[r].significant &&
		   kmahip_res_line(name, &rows[r], cover[t], aln_len[t], depth[t], 1.0, 0.0, line, (1 << 16) + 512)) fputs(line, stdout);
	}
	fclose(names);
	fflush(stdout);
	kmahip_ws_destroy(ws);
	kmahip_db_close(db);
	kmahip_ingest_close(ing);
	return 0;
}
