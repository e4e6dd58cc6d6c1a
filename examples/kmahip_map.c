/* kmahip_map.c -- `kma -i reads.fq[.gz] -o out -t_db db -1t1` (or `-ipe r1.fq r2.fq ... -apm p -1t1`, or `-i reads.fq -Mt1 n [-bcNano]`)
 * on an MI355X without the reference: plain C99 over the C-ABI of libkmahip.so. Writes out.res, out.fsa and out.frag.gz, byte
 * for byte what KMA 1.5.1 writes with one thread (the .gz after decompression).
 *
 *     kmahip_map -i reads.fq.gz -t_db db -o out                     (the reference's default mode: chain finder, reads may map in pieces)
 *     kmahip_map -i reads.fq.gz -t_db db -o out -1t1
 *     kmahip_map -ipe r1.fq.gz r2.fq.gz -t_db db -o out -1t1
 *     kmahip_map -i ont.fq.gz -t_db db -o out -Mt1 1 -bcNano        (every read against template 1, runKMA_Mt1 mt1.c:86-500)
 *
 * Stage 1 (kmahip_ingest_*: parse, trim with KMA's defaults, pack), the whole device run in one call (kmahip_run_se: stage 2,
 * 3a, ConClave, `.res` statistics, traceback, pile-up, consensus), then the three writers.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>
#include <sys/stat.h>
#include <unistd.h>

#include "kmahip.h"

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* stage 1 on a thread of its own, beside HIP start-up and the loading of the index */
typedef struct ingest_job { const char *in1, *in2; kmahip_ingest *ing; kmahip_read_batch b; int rc; char err[512]; double t_done; } ingest_job;
static void *ingest_main(void *arg) {
	ingest_job *j = (ingest_job *) arg;
	j->rc = kmahip_ingest_open(j->in1, j->in2, NULL, &j->ing);
	if(!j->rc) j->rc = kmahip_ingest_next(j->ing, INT64_MAX, &j->b);
	if(j->rc) { strncpy(j->err, kmahip_last_error(), sizeof j->err - 1); j->err[sizeof j->err - 1] = 0; }
	j->t_done = now_s();
	return NULL;
}

/* The per-read result columns (half a gigabyte for 10 M reads) come from calloc: their pages do not exist until something writes
 * them. A thread writes them -- an atomic OR of zero per page, which changes nothing whatever the run has already copied there --
 * while the GPU is busy with stages 2 and 3a, so that the copy at the end of the run finds the pages in place. */
typedef struct touch_job { char *p[4]; size_t n[4]; } touch_job;
static void *touch_main(void *arg) {
	touch_job *j = (touch_job *) arg;
	for(int a = 0; a < 4; ++a) for(size_t i = 0; i < j->n[a]; i += 4096) __atomic_fetch_or(&j->p[a][i], 0, __ATOMIC_RELAXED);
	return NULL;
}

/* While stage 1 is still reading: one run on reads cut out of the first template (word-aligned windows of 150 bases, as many reads as the
 * traceback keeps lanes), so that what the device path pays once per process -- the first launch of every kernel, the scratch that does
 * not depend on the batch -- is paid beside the I/O and not after it. The results are thrown away. */
static void warm_up(kmahip_db *db, kmahip_ws *ws, const char *prefix, int64_t D, const kmahip_params *par) {
	char path[4096];
	int32_t hdr[3] = {0, 0, 0};
	uint64_t w[64];
	snprintf(path, sizeof path, "%s.length.b", prefix);
	FILE *f = fopen(path, "rb");
	if(!f || fread(hdr, 4, 3, f) != 3) { if(f) fclose(f); return; }
	fclose(f);
	const int tl = hdr[2];
	if(tl < 192) return;
	const int tw = (tl >> 5) + 1 < 64 ? (tl >> 5) + 1 : 64;
	snprintf(path, sizeof path, "%s.seq.b", prefix);
	f = fopen(path, "rb");
	if(!f || fread(w, 8, (size_t) tw, f) != (size_t) tw) { if(f) fclose(f); return; }
	fclose(f);
	const int64_t n = 262144, nwin = tw - 5;
	uint64_t *seq = calloc((size_t) n * 6 + 2, 8);
	int64_t *off = malloc((size_t) (n + 1) * 8), *noff = calloc((size_t) n + 1, 8);
	int32_t *len = malloc((size_t) n * 4), none = 0;
	kmahip_run run;
	memset(&run, 0, sizeof run);
	run.rows = calloc((size_t) D, sizeof *run.rows); run.rows_cap = D;
	run.assembly.cover = calloc((size_t) D, 8); run.assembly.aln_len = calloc((size_t) D, 8);
	run.assembly.depth = calloc((size_t) D, 8); run.assembly.asm_len = calloc((size_t) D, 8);
	if(seq && off && noff && len && run.rows && run.assembly.cover && run.assembly.aln_len && run.assembly.depth && run.assembly.asm_len && nwin > 0) {
		for(int64_t i = 0; i < n; ++i) {
			const uint64_t *src = w + i % nwin;
			for(int x = 0; x < 4; ++x) seq[6 * i + x] = src[x];
			seq[6 * i + 4] = src[4] & (~0ull << (64 - 2 * 22));          /* 150 = 4 x 32 + 22 bases */
			off[i] = 6 * i; len[i] = 150;
		}
		off[n] = 6 * n;
		kmahip_reads r;
		memset(&r, 0, sizeof r);
		r.n_reads = n; r.seq = seq; r.seq_off = off; r.len = len; r.N = &none; r.N_off = noff; r.seq_words = 6 * n; r.N_total = 0; r.max_len = 150;
		(void) kmahip_run_se(db, ws, &r, par, 0.05, 1, 0, &run);
	}
	free(seq); free(off); free(noff); free(len); free(run.rows);
	free(run.assembly.cover); free(run.assembly.aln_len); free(run.assembly.depth); free(run.assembly.asm_len);
}

static void die(const char *what) { fprintf(stderr, "kmahip_map: %s: %s\n", what, kmahip_last_error()); exit(1); }
static void *xcalloc(size_t n, size_t sz) { void *p = calloc(n ? n : 1, sz); if(!p) { fprintf(stderr, "kmahip_map: out of memory\n"); exit(1); } return p; }

int main(int argc, char **argv) {
	const char *prefix = NULL, *input = NULL, *input2 = NULL, *out = NULL;
	int mt1 = 0, bc_nano = 0, one2one = 0, chain = 0;
	long long max_frag = 0;
	for(int a = 1; a < argc; ++a) {
		if(!strcmp(argv[a], "-Mt1") && a + 1 < argc) { mt1 = atoi(argv[++a]); continue; }
		if(!strcmp(argv[a], "-bcNano")) { bc_nano = 1; continue; }
		if(!strcmp(argv[a], "-1t1")) { one2one = 1; continue; }
		if(!strcmp(argv[a], "-chain")) { chain = 1; continue; }      /* (same as leaving -1t1 out) */
		if(!strcmp(argv[a], "-t_db") && a + 1 < argc) prefix = argv[++a];
		else if(!strcmp(argv[a], "-i") && a + 1 < argc) input = argv[++a];
		else if(!strcmp(argv[a], "-ipe") && a + 2 < argc) { input = argv[++a]; input2 = argv[++a]; }
		else if(!strcmp(argv[a], "-o") && a + 1 < argc) out = argv[++a];
		else if(!strcmp(argv[a], "-mf") && a + 1 < argc) max_frag = atoll(argv[++a]);       /* fragments per assembly chunk (kma.c:1045) */
		else { fprintf(stderr, "usage: kmahip_map (-i reads.fq[.gz] | -ipe r1.fq[.gz] r2.fq[.gz]) -t_db <index prefix> -o <output prefix> [-1t1 | -chain] [-mf <fragments per chunk>] [-Mt1 <template> [-bcNano]]\n"); return 2; }
	}
	if(!prefix || !input || !out) { fprintf(stderr, "kmahip_map: -i, -t_db and -o are required\n"); return 2; }
	/* like the reference: without -1t1 (and without -Mt1) the template finder is save_kmers_chain, reads may map in pieces */
	if(!one2one && !mt1) chain = 1;
	if(chain && input2) { fprintf(stderr, "kmahip_map: paired input needs -1t1 (the default mode is built for single-end input)\n"); return 2; }

	const double t_start = now_s();
	/* stage 1: the whole file as one batch (the arrays stay owned by the reader), while the device and the index come up */
	ingest_job job;
	memset(&job, 0, sizeof job);
	job.in1 = input; job.in2 = input2;
	pthread_t ingest_thread;
	if(pthread_create(&ingest_thread, NULL, ingest_main, &job)) { fprintf(stderr, "kmahip_map: cannot start a thread\n"); return 1; }

	kmahip_db *db; kmahip_ws *ws; kmahip_params par; kmahip_db_info info;
	if(kmahip_init(0) || kmahip_db_open(prefix, &db) || kmahip_ws_create(db, &ws) || kmahip_db_get_info(db, &info)) die("open");
	kmahip_default_params(&par);
	const int64_t D = info.DB_size;
	const double t_open = now_s();
	{	/* (worth it when stage 1 still has a few hundred milliseconds of reading in front of it: a gigabyte of text, or 128 MB of .gz;
		 * behind a shorter input the warm-up itself would be what the run waits for) */
		struct stat sb;
		const size_t il = strlen(input);
		const int gz = il > 3 && !strcmp(input + il - 3, ".gz");
		if(!mt1 && !getenv("KMAHIP_MAP_NO_WARMUP") && stat(input, &sb) == 0 && sb.st_size >= (gz ? (128ll << 20) : (1ll << 30))) warm_up(db, ws, prefix, D, &par);
	}
	pthread_join(ingest_thread, NULL);
	if(job.rc) { fprintf(stderr, "kmahip_map: ingest: %s\n", job.err); return 1; }
	kmahip_ingest *ing = job.ing;
	const kmahip_read_batch b = job.b;
	const int64_t n = b.reads.n_reads;
	const double t_ingest = now_s();

	/* everything on the device, one call */
	int64_t tbases = 0;
	{	/* consensus capacity: template bases (from <prefix>.length.b: DB_size ints, the first is the k-mer index size) + slack */
		char path[4096];
		snprintf(path, sizeof path, "%s.length.b", prefix);
		FILE *f = fopen(path, "rb");
		int32_t v, i = 0;
		if(!f) { fprintf(stderr, "kmahip_map: cannot open %s\n", path); return 1; }
		while(fread(&v, 4, 1, f) == 1) { if(i >= 2 && i <= D) tbases += v; ++i; }      /* [count][length of entry 0 = k][template 1] ... [template D - 1] */
		fclose(f);
	}
	kmahip_run run;
	memset(&run, 0, sizeof run);
	run.rows = xcalloc((size_t) D, sizeof *run.rows); run.rows_cap = D;
	run.assembly.cover = xcalloc((size_t) D, 8); run.assembly.aln_len = xcalloc((size_t) D, 8);
	run.assembly.depth = xcalloc((size_t) D, 8); run.assembly.asm_len = xcalloc((size_t) D, 8);
	run.assembly.consensus_cap = 4 * tbases + 4 * D + (1 << 20);      /* template columns + insertion columns + a NUL each */
	run.assembly.consensus = xcalloc((size_t) run.assembly.consensus_cap, 1);
	run.assembly.consensus_off = xcalloc((size_t) D, 8);
	for(int64_t t = 0; t < D; ++t) run.assembly.consensus_off[t] = -1;
	run.tmpl = xcalloc((size_t) n + 1, 4); run.n_hits = xcalloc((size_t) n + 1, 4); run.rc = xcalloc((size_t) n + 1, 4);
	run.trace_stats = xcalloc((size_t) n * 10 + 10, 4);
	run.caller = bc_nano; run.sig90 = bc_nano;      /* -bcNano (kma.c:762-766) */
	touch_job tj = { { (char *) run.tmpl, (char *) run.n_hits, (char *) run.rc, (char *) run.trace_stats },
	                 { ((size_t) n + 1) * 4, ((size_t) n + 1) * 4, ((size_t) n + 1) * 4, ((size_t) n * 10 + 10) * 4 } };
	pthread_t touch_thread;
	const int touching = !input2 && !chain && n > 100000 && !getenv("KMAHIP_MAP_NO_TOUCH") && !pthread_create(&touch_thread, NULL, touch_main, &tj);
	char fpath[4096];
	snprintf(fpath, sizeof fpath, "%s.frag.gz", out);
	if(mt1) {
		kmahip_assemble_opts ao;
		memset(&ao, 0, sizeof ao);
		ao.evalue = 0.05; ao.bcd = 1; ao.order = 1; ao.caller = bc_nano; ao.sig90 = bc_nano;
		if(input2) { fprintf(stderr, "kmahip_map: -Mt1 with -ipe is not supported\n"); return 2; }
		if(kmahip_run_mt1(db, ws, &b.reads, mt1, one2one, &par, &ao, &run)) die("kmahip_run_mt1");
	} else if(chain) {
		if(input2) { fprintf(stderr, "kmahip_map: -chain with -ipe is not supported\n"); return 2; }
		if(kmahip_run_chain(db, ws, &b.reads, b.names, b.name_off, &par, NULL, 0.05, 1, max_frag, fpath, &run)) die("kmahip_run_chain");
	} else if(input2) { if(kmahip_run_pe(db, ws, &b, &par, 0.05, 1, max_frag, fpath, &run)) die("kmahip_run_pe"); }
	else if(kmahip_run_se(db, ws, &b.reads, &par, 0.05, 1, max_frag, &run)) die("kmahip_run_se");

	if(touching) pthread_join(touch_thread, NULL);
	const double t_run = now_s();
	/* out.res + out.fsa: names from <prefix>.name, one per line, in template order */
	char path[4096], *name = xcalloc(1 << 16, 1), *line = xcalloc((1 << 16) + 512, 1);
	snprintf(path, sizeof path, "%s.name", prefix);
	FILE *names = fopen(path, "r");
	snprintf(path, sizeof path, "%s.res", out);
	FILE *res = fopen(path, "w");
	snprintf(path, sizeof path, "%s.fsa", out);
	FILE *fsa = fopen(path, "w");
	if(!names || !res || !fsa) { fprintf(stderr, "kmahip_map: cannot open the name file or the outputs\n"); return 1; }
	fputs("#Template\tScore\tExpected\tTemplate_length\tTemplate_Identity\tTemplate_Coverage\tQuery_Identity\tQuery_Coverage\tDepth\tq_value\tp_value\n", res);
	int64_t r = 0;
	size_t fsa_cap = 1 << 16;
	char *fsa_buf = xcalloc(fsa_cap, 1);
	for(int64_t t = 1; t < D && fgets(name, 1 << 16, names); ++t) {
		name[strcspn(name, "\n")] = 0;
		while(r < run.n_rows && run.rows[r].template_id < t) ++r;
		if(!(r < run.n_rows && run.rows[r].template_id == t && run.rows[r].significant)) continue;
		if(!kmahip_res_line(name, &run.rows[r], run.assembly.cover[t], run.assembly.aln_len[t], run.assembly.depth[t], 1.0, 0.0, line, (1 << 16) + 512)) continue;
		fputs(line, res);
		/* printConsensus (printconsensus.c:38-60): the consensus line without its '-' columns, 60 per line */
		fprintf(fsa, ">%s\n", name);
		const char *c = run.assembly.consensus + run.assembly.consensus_off[t];
		const size_t clen = strlen(c);
		if(clen + clen / 60 + 2 > fsa_cap) { fsa_cap = 2 * (clen + clen / 60 + 2); free(fsa_buf); fsa_buf = xcalloc(fsa_cap, 1); }
		char *o = fsa_buf;
		int col = 0;
		for(; *c; ++c) if(*c != '-') { *o++ = *c; if(++col == 60) { *o++ = '\n'; col = 0; } }
		if(col) *o++ = '\n';
		fwrite(fsa_buf, 1, (size_t) (o - fsa_buf), fsa);
	}
	fclose(names); fclose(res); fclose(fsa);

	const double t_res = now_s();
	/* out.frag.gz (the paired run has written it itself: its fragments are in record order, not read order) */
	int64_t frag_rows = 0;
	if(!input2 && !chain && kmahip_frag_write2(fpath, db, &b.reads, run.rc, run.tmpl, run.n_hits, run.trace_stats, max_frag, mt1 ? 1 : 0, b.names, b.name_off, &frag_rows)) die("kmahip_frag_write");
	const double t_frag = now_s();
	fprintf(stderr, "# kmahip_map: %lld reads, %lld fragment rows; wall: ingest %.2f s beside open %.2f (both done after %.2f), device run %.2f, .res + .fsa %.2f, .frag.gz %.2f | "
	        "upload %.1f ms, stages 2+3a %.1f, ConClave %.1f, traceback %.1f, pile-up + consensus %.1f, columns back %.1f%s\n", (long long) n, (long long) frag_rows,
	        job.t_done - t_start, t_open - t_start, t_ingest - t_start, t_run - t_ingest, t_res - t_run, t_frag - t_res, run.ms[0], run.ms[1], run.ms[2], run.ms[3], run.ms[4],
	        input2 || chain ? 0.0 : run.ms[5],
	        input2 || chain ? " (the device run wrote the .frag.gz)" : "");
	/* every output is closed; the process ends here instead of unmapping gigabytes of reads and scratch one by one
	 * (KMAHIP_MAP_TEARDOWN=1: release everything in order, e.g. under a leak checker) */
	if(getenv("KMAHIP_MAP_TEARDOWN")) {
		kmahip_ws_destroy(ws);
		kmahip_db_close(db);
		kmahip_ingest_close(ing);
		return 0;
	}
	fflush(NULL);
	_exit(0);
}
