/* kmahip_map.c -- `kma -i reads.fq[.gz] -o out -t_db db [-1t1]` (or `-ipe r1.fq r2.fq ... -apm p -1t1`, or `-i reads.fq -Mt1 n [-bcNano]`)
 * on MI355X without the reference: plain C99 over the C-ABI of libkmahip.so. Writes out.res, out.fsa, out.aln and out.frag.gz, byte for
 * byte what KMA 1.5.1 writes with one thread (the .gz after decompression).
 *
 *     kmahip_map -i reads.fq.gz -t_db db -o out                     (the reference's default mode: chain finder, reads may map in pieces)
 *     kmahip_map -i reads.fq.gz -t_db db -o out -1t1
 *     kmahip_map -ipe r1.fq.gz r2.fq.gz -t_db db -o out -1t1 -apm p
 *     kmahip_map -ipe r1.fq.gz r2.fq.gz -t_db db -o out              (paired input in the default mode: couples by union, single records in pieces)
 *     kmahip_map -i ont.fq.gz -t_db db -o out -Mt1 1 -bcNano        (every read against template 1, runKMA_Mt1 mt1.c:86-500)
 *     kmahip_map -gpus 8 -i reads.fq -t_db db -o out -1t1           (one process per GPU, the reads sharded; see below)
 *
 * The command line is the reference's (kma.c:351-1248) for the options this path implements; anything else is refused, loudly.
 * Stage 1 (kmahip_ingest_*: parse, trim, pack), the whole device run in one call (kmahip_run_se / _pe / _chain / _mt1: stage 2, 3a,
 * ConClave, `.res` statistics, traceback, pile-up, consensus), then the three writers.
 *
 * Several GPUs of one node (every mode above): `-gpus N` starts N copies of this program, one per device, before anything
 * touches a GPU; each copy is a rank (KMAHIP_RANK / KMAHIP_WORLD / KMAHIP_KEY in its environment; under another launcher RANK /
 * WORLD_SIZE / LOCAL_RANK and MASTER_PORT are read instead). A rank parses its byte range of the FASTQ, maps its reads, and
 * kmahip_run_se_sharded / _pe_sharded / _chain_sharded / _mt1_sharded do the exchanges (kmahip.h) -- over RCCL, or staged through shared memory with
 * KMAHIP_COMM=shm (KMAHIP_SHARE_GPU=1 puts every rank on device 0: a rehearsal on a one-GPU box).
 */
#define _POSIX_C_SOURCE 200809L
#define _DEFAULT_SOURCE
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include "kmahip.h"

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* seconds since the kernel started this process (exec, loader and libraries included): /proc/self/stat field 22 against the boot clock */
static double since_process_start(void) {
	FILE *f = fopen("/proc/self/stat", "r");
	char buf[2048];
	if(!f) return -1;
	const size_t got = fread(buf, 1, sizeof buf - 1, f);
	fclose(f);
	buf[got] = 0;
	const char *p = strrchr(buf, ')');          /* behind the command name, which may hold anything */
	if(!p) return -1;
	unsigned long long start = 0;
	int field = 2;
	for(p += 1; *p && field < 22; ++p) if(*p == ' ') { ++field; if(field == 22) { start = strtoull(p + 1, NULL, 10); break; } }
	struct timespec ts;
	clock_gettime(CLOCK_BOOTTIME, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec - (double) start / (double) sysconf(_SC_CLK_TCK);
}

/* peak resident set of this process so far, MB (VmHWM of /proc/self/status) */
static double peak_rss_mb(void) {
	FILE *f = fopen("/proc/self/status", "r");
	char line[256];
	double mb = -1;
	if(!f) return mb;
	while(fgets(line, sizeof line, f)) if(!strncmp(line, "VmHWM:", 6)) { mb = strtod(line + 6, NULL) / 1024.0; break; }
	fclose(f);
	return mb;
}

/* -Mt1 with paired input (printFsa_pairMt1, mt1.c:61-83): the mates of a couple are two records of their own, the second one reverse
 * complemented (strrc on the bases: an N stays an N) -- done here on the packed batch, in the reader's own arrays: 32 bases per word from
 * the top bits down, N packed as A with its position listed, the bits behind the last base zero */
static void rc_second_mates(kmahip_read_batch *b) {
	uint64_t *seq = (uint64_t *) b->reads.seq;
	int32_t *N = (int32_t *) b->reads.N;
	for(int64_t r = 0; r < b->reads.n_reads; ++r) {
		if(b->pair[r] != 2) continue;
		uint64_t *w = seq + b->reads.seq_off[r];
		const int L = b->reads.len[r];
		int32_t *n = N + b->reads.N_off[r];
		const int nn = (int) (b->reads.N_off[r + 1] - b->reads.N_off[r]);
		for(int i = 0, j = L - 1; i < j; ++i, --j) {          /* swap and complement base by base (reads are short here; this is not the hot path) */
			const int si = 62 - ((i & 31) << 1), sj = 62 - ((j & 31) << 1);
			const uint64_t bi = (w[i >> 5] >> si) & 3u, bj = (w[j >> 5] >> sj) & 3u;
			w[i >> 5] = (w[i >> 5] & ~(3ull << si)) | ((3u - bj) << si);
			w[j >> 5] = (w[j >> 5] & ~(3ull << sj)) | ((3u - bi) << sj);
		}
		if(L & 1) { const int m = L >> 1, sm = 62 - ((m & 31) << 1); w[m >> 5] ^= 3ull << sm; }
		for(int x = 0, y = nn - 1; x <= y; ++x, --y) { const int32_t a = L - 1 - n[y], c = L - 1 - n[x]; n[x] = a; n[y] = c; }
		for(int x = 0; x < nn; ++x) { const int q = n[x], sq = 62 - ((q & 31) << 1); w[q >> 5] &= ~(3ull << sq); }      /* N packed as A */
	}
}

/* stage 1 on a thread of its own, beside HIP start-up and the loading of the index */
typedef struct ingest_job {
	const char *in1, *in2; int rc_mates;
	kmahip_trim trim;
	int part, parts, whole_input;
	kmahip_ingest *ing; kmahip_read_batch b; int rc; char err[512]; double t_done;
} ingest_job;
static void *ingest_main(void *arg) {
	ingest_job *j = (ingest_job *) arg;
	j->rc = kmahip_ingest_open_part(j->in1, j->in2, &j->trim, j->part, j->parts, &j->ing, &j->whole_input);
	if(!j->rc) j->rc = kmahip_ingest_next(j->ing, INT64_MAX, &j->b);
	if(!j->rc && j->rc_mates) rc_second_mates(&j->b);
	/* the whole input in one batch: an input that breaks off (a truncated .gz, a record that is no FASTQ) delivers what came before
	 * it; whether it did has to be asked, or the run would end well on half the reads */
	if(!j->rc) j->rc = kmahip_ingest_status(j->ing);
	if(j->rc && !j->err[0]) { strncpy(j->err, kmahip_last_error(), sizeof j->err - 1); j->err[sizeof j->err - 1] = 0; }
	j->t_done = now_s();
	return NULL;
}

/* stage 1 batch by batch for the session (single end, -1t1, one rank): the reader parses the next batch while the device works on the
 * one before; a batch is handed over (state 1), uploaded by the main thread, and given back (state 0) */
typedef struct stream_job {
	const char *in1, *in2; int rc_mates;
	char **list1, **list2; int n_files;      /* (more than one input file: read one after the other, kma.c:370-460 run_input*) */
	kmahip_trim trim;
	int64_t batch_reads, batch_bases;
	kmahip_ingest *ing; kmahip_read_batch b;
	int state, rc; char err[512]; double t_done;
	pthread_mutex_t mu; pthread_cond_t cv;
} stream_job;
static void *stream_main(void *arg) {
	stream_job *j = (stream_job *) arg;
	int whole = 0, file = 0;
	int rc = kmahip_ingest_open_part(j->in1, j->in2, &j->trim, 0, 1, &j->ing, &whole);
	if(!rc && j->batch_bases > 0) rc = kmahip_ingest_set_batch_bases(j->ing, j->batch_bases);
	for(;;) {
		if(!rc) rc = kmahip_ingest_next(j->ing, j->batch_reads, &j->b);
		if(!rc && j->rc_mates) rc_second_mates(&j->b);
		int end = rc || j->b.reads.n_reads == 0;
		if(end && !rc) rc = kmahip_ingest_status(j->ing);
		if(end && !rc && file + 1 < j->n_files) {
			/* the next file of the list (its own phred scale, like the reference's loop over the files; a batch never spans two files).
			 * The batch handed over last has been given back (state 0), so nothing points into this reader any more */
			++file;
			kmahip_ingest_close(j->ing); j->ing = NULL;
			rc = kmahip_ingest_open_part(j->list1[file], j->in2 ? (j->in2[0] ? j->list2[file] : "") : NULL, &j->trim, 0, 1, &j->ing, &whole);
			if(!rc && j->batch_bases > 0) rc = kmahip_ingest_set_batch_bases(j->ing, j->batch_bases);
			if(!rc) continue;
			end = 1;
		}
		pthread_mutex_lock(&j->mu);
		if(end) {
			j->rc = rc;
			if(rc) { strncpy(j->err, kmahip_last_error(), sizeof j->err - 1); j->err[sizeof j->err - 1] = 0; }
			j->t_done = now_s();
			j->state = 2;
			pthread_cond_broadcast(&j->cv);
			pthread_mutex_unlock(&j->mu);
			return NULL;
		}
		j->state = 1;
		pthread_cond_broadcast(&j->cv);
		while(j->state != 0) pthread_cond_wait(&j->cv, &j->mu);
		pthread_mutex_unlock(&j->mu);
	}
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

/* the pipe to the waiting parent (see main): one byte = the exit status, written when the outputs are complete or the run has failed */
static int g_done_fd = -1;
static void finish(int status) {
	fflush(NULL);
	if(g_done_fd >= 0) {
		const unsigned char st = (unsigned char) status;
		if(write(g_done_fd, &st, 1) != 1) { /* the parent is gone */ }
		close(g_done_fd); g_done_fd = -1;
		close(STDOUT_FILENO); close(STDERR_FILENO);          /* (whoever reads our output through a pipe sees its end now, not after the teardown) */
	}
	_exit(status);
}
/* (a reader thread may be inflating, workers of the library may be running: no destructors, no atexit handlers on the way out) */
static void die(const char *what) { fprintf(stderr, "kmahip_map: %s: %s\n", what, kmahip_last_error()); finish(1); }
static void fail(const char *msg) { fprintf(stderr, "kmahip_map: %s\n", msg); finish(1); }
static void *xcalloc(size_t n, size_t sz) { void *p = calloc(n ? n : 1, sz); if(!p) fail("out of memory"); return p; }

static void usage(void) {
	fprintf(stderr, "usage: kmahip_map (-i reads.fq[.gz] | -ipe r1.fq[.gz] r2.fq[.gz] [-apm p|u] | -int interleaved.fq[.gz] [-apm p|u]) -t_db <index prefix> -o <output prefix> [-1t1] [-Mt1 <template>] [-bcNano] [-bc90] [-bc <support>] [-bcg] [-ref_fsa [n]] [-dense]\n"
	                "       [-t threads] [-nc] [-na] [-nf] [-mf fragments] [-ml len] [-xl len] [-mp phred] [-mi phred] [-eq q] [-mq q] [-ts bases] [-mrs f] [-mrc f] [-mct f]\n"
	                "       [-e evalue] [-bcd depth] [-ID id] [-md depth] [-ex_mode] [-gpus N]\n"
	                "       [-reward n] [-gapopen n] [-gapextend n] [-localopen n] [-Npenalty n] [-per n] [-transition n] [-transversion n] [-penalty n] [-cge]\n"
	                "(the options of kma 1.5.1 this path implements; -apm takes p or u; everything else is refused)\n");
}

static long long need_int(int argc, char **argv, int *a, const char *what) {
	char *end;
	if(*a + 1 >= argc) { fprintf(stderr, "kmahip_map: %s needs a value\n", what); exit(2); }
	const long long v = strtoll(argv[++*a], &end, 10);
	if(*end) { fprintf(stderr, "kmahip_map: invalid argument at \"%s\"\n", what); exit(1); }
	return v;
}
static double need_num(int argc, char **argv, int *a, const char *what) {
	char *end;
	if(*a + 1 >= argc) { fprintf(stderr, "kmahip_map: %s needs a value\n", what); exit(2); }
	const double v = strtod(argv[++*a], &end);
	if(*end) { fprintf(stderr, "kmahip_map: invalid argument at \"%s\"\n", what); exit(1); }
	return v;
}

/* a profiler's tool library loaded into this process (rocprofv3 sets ROCP_TOOL_LIBRARIES and preloads librocprofiler-sdk-tool; the
 * older tools set HSA_TOOLS_LIB) initialises the GPU before main() runs. An LD_PRELOAD of anything else (an allocator, a sandbox's
 * guard library) does not count. */
static int tool_preloaded(void) {
	const char *v[] = {"ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD"};
	for(size_t i = 0; i < sizeof v / sizeof *v; ++i) { const char *e = getenv(v[i]); if(e && *e) return 1; }
	const char *pre = getenv("LD_PRELOAD");
	return pre && (strstr(pre, "rocprof") || strstr(pre, "roctracer") || strstr(pre, "roctx")) ? 1 : 0;
}

/* `-gpus N`: N copies of this program, one per device, started (fork + exec of this binary) BEFORE this process has touched a GPU;
 * the exit status is the first non-zero one of the copies, and when a copy fails the others are ended (they would wait for it in
 * the communicator until its timeout). Never behind a profiler: there the GPU is initialised before main(), and replacing a
 * GPU-initialised process by exec is what must not happen -- profile one rank instead (tools/README.md). */
static int launch_ranks(int gpus, char **argv) {
	if(tool_preloaded()) {
		fprintf(stderr, "kmahip_map: -gpus starts its ranks by fork + exec, which a process with a profiler's tool library loaded (ROCP_TOOL_LIBRARIES / HSA_TOOLS_LIB / "
		                "a rocprof library in LD_PRELOAD) must not do. Profile one rank directly: KMAHIP_RANK=r KMAHIP_WORLD=N KMAHIP_KEY=<key> <profiler> -- kmahip_map ... (without -gpus), the other ranks started the same way\n");
		return 2;
	}
	char key[64], val[32];
	snprintf(key, sizeof key, "%ld_%ld", (long) getpid(), (long) time(NULL));
	setenv("KMAHIP_KEY", key, 1);
	snprintf(val, sizeof val, "%d", gpus);
	setenv("KMAHIP_WORLD", val, 1);
	pid_t *pid = calloc((size_t) gpus, sizeof *pid);
	if(!pid) return 1;
	int started = 0, status = 0;
	for(int r = 0; r < gpus; ++r) {
		pid[r] = fork();
		if(pid[r] < 0) { perror("kmahip_map: fork"); status = 1; break; }
		if(pid[r] == 0) {
			snprintf(val, sizeof val, "%d", r);
			setenv("KMAHIP_RANK", val, 1);
			execv("/proc/self/exe", argv);
			perror("kmahip_map: exec");
			_exit(127);
		}
		++started;
	}
	for(int left = started; left > 0; --left) {
		int st = 0;
		const pid_t w = status ? -1 : wait(&st);
		if(status || w < 0 || !WIFEXITED(st) || WEXITSTATUS(st)) {
			if(!status) status = w >= 0 && WIFEXITED(st) && WEXITSTATUS(st) ? WEXITSTATUS(st) : 1;
			for(int r = 0; r < started; ++r) if(pid[r] > 0 && pid[r] != w) kill(pid[r], SIGTERM);      /* (exactly the ranks started here) */
			for(int r = 0; r < started; ++r) if(pid[r] > 0 && pid[r] != w) waitpid(pid[r], NULL, 0);
			break;
		}
		for(int r = 0; r < started; ++r) if(pid[r] == w) pid[r] = 0;
	}
	free(pid);
	return status;
}

int main(int argc, char **argv) {
	const char *prefix = NULL, *input = NULL, *input2 = NULL, *out = NULL;
	int Ts = -2, Tv = -2;          /* -transition / -transversion (kma.c:335-336) */
	int cmp_mode = 0, lc = 0, mem_mode = 0;      /* -and / -oa, -lc, -mem_mode */
	int pm = 0, fpm = 0;           /* -pm / -fpm (1 p, 2 u; 0: not given) */
	char *list1[256], *list2[256]; int n_files = 0;          /* the input files (mate files side by side) */
	int mt1 = 0, one2one = 0, chain = 0, apm = 0, no_cons = 0, no_frag = 0, no_aln = 0, gpus = 0, threads = 0, bcd = 1;
	int base_call = 0, sig_mode = 0, ref_fsa = 0, dense = 0;      /* as kmahip_assemble_opts.caller (0-2 here) / .sig90; printconsensus.c's ref_fsa */
	double support = 0;
	long long max_frag = 0;
	double evalue = 0.05, ID_t = 1.0, Depth_t = 0.0;
	kmahip_params par;
	kmahip_chain_params cp;
	kmahip_trim trim;
	kmahip_default_params(&par);
	kmahip_trim_default(&trim);
	cp.minlen = 16; cp.pad_ = 0; cp.coverT = 0.1; cp.mrs = 0.5;
	for(int a = 1; a < argc; ++a) {
		const char *o = argv[a];
		if(!strcmp(o, "-Mt1")) mt1 = (int) need_int(argc, argv, &a, o);                        /* kma.c:923 */
		else if(!strcmp(o, "-bcNano")) { if(sig_mode == 0) sig_mode = 1; base_call = 1; }   /* kma.c:762-766 */
		else if(!strcmp(o, "-dense")) dense = 1;                                               /* kma.c:662: alnToMatDense */
		else if(!strcmp(o, "-bc90")) sig_mode = 1;                                              /* kma.c:758 */
		else if(!strcmp(o, "-bcg")) base_call = 2;                                              /* kma.c:760: orgBaseCaller */
		else if(!strcmp(o, "-bc")) {                                                            /* kma.c:744-757: with a value the support a call needs, without one back to significantNuc */
			if(a + 1 < argc && argv[a + 1][0] != '-') {
				support = need_num(argc, argv, &a, o);
				if(support < 0 || 1 < support) { fprintf(stderr, "kmahip_map: invalid argument at \"-bc\"\n"); return 1; }
				sig_mode = 2;
			} else sig_mode = 0;
		}
		else if(!strcmp(o, "-ref_fsa")) {                                                       /* kma.c:671-684 */
			ref_fsa = 1;
			if(a + 1 < argc && argv[a + 1][0] != '-') { ref_fsa = (int) need_int(argc, argv, &a, o); if(ref_fsa == 0) ref_fsa = 2; }
		}
		else if(!strcmp(o, "-1t1")) one2one = 1;                                                /* kma.c:686 */
		else if(!strcmp(o, "-chain")) chain = 1;                                                /* (our own: the same as leaving -1t1 out) */
		else if(!strcmp(o, "-t_db") && a + 1 < argc) prefix = argv[++a];
		else if(!strcmp(o, "-i") || !strcmp(o, "-int") || !strcmp(o, "-ipe")) {                 /* kma.c:371-435: each takes a list of files */
			const int pe = o[2] == 'p';
			if(n_files) { fprintf(stderr, "kmahip_map: one of -i, -ipe and -int, once\n"); return 2; }
			while(a + 1 < argc && argv[a + 1][0] != '-' && n_files < 256) {
				list1[n_files] = argv[++a]; list2[n_files] = NULL;
				if(pe) {
					if(!(a + 1 < argc && argv[a + 1][0] != '-')) { fprintf(stderr, "Uneven number of paired end files.\n"); return 1; }
					list2[n_files] = argv[++a];
				}
				++n_files;
			}
			if(!n_files) { fprintf(stderr, "kmahip_map: %s needs a file\n", o); return 1; }
			input = list1[0]; input2 = pe ? list2[0] : (o[2] == 'n' ? "" : NULL);      /* interleaved: the paired reader on one file (kmahip_ingest_open_part with path2 = "") */
		}
		else if(!strcmp(o, "-o") && a + 1 < argc) out = argv[++a];
		else if(!strcmp(o, "-apm")) {                                                           /* kma.c:472: p, u or f */
			if(a + 1 >= argc || (argv[a + 1][0] != 'p' && argv[a + 1][0] != 'u' && argv[a + 1][0] != 'f')) { fprintf(stderr, "kmahip_map: -apm takes p (pairing reward, save_kmers_penaltyPair / alnFragsPenaltyPE), u (union, save_kmers_unionPair / alnFragsUnionPE) ; f (forced) is read and refused below\n"); return 1; }
			apm = argv[++a][0] == 'p' ? 1 : (argv[a][0] == 'u' ? 2 : 3);
			pm = fpm = 0;
		}
		else if(!strcmp(o, "-pm") || !strcmp(o, "-fpm")) {                                      /* kma.c:437-465: the two stages apart */
			if(a + 1 >= argc || (argv[a + 1][0] != 'p' && argv[a + 1][0] != 'u' && argv[a + 1][0] != 'f')) { fprintf(stderr, "kmahip_map: %s takes p or u (f is read and refused below)\n", o); return 1; }
			const int v = argv[++a][0] == 'p' ? 1 : (argv[a][0] == 'u' ? 2 : 3);
			if(o[1] == 'p') pm = v; else fpm = v;
		}
		else if(!strcmp(o, "-t")) {                                                             /* kma.c:529: a value is optional */
			if(a + 1 < argc && argv[a + 1][0] != '-') threads = (int) need_int(argc, argv, &a, o);
			if(threads < 1) threads = 1;
		}
		else if(!strcmp(o, "-nc")) no_cons = 1;                                                 /* kma.c:1018-1022 */
		else if(!strcmp(o, "-na")) no_aln = 1;                                                  /* kma.c:1023: no .aln file */
		else if(!strcmp(o, "-nf")) no_frag = 1;
		else if(!strcmp(o, "-mf")) { max_frag = need_int(argc, argv, &a, o); if(max_frag < 0) { fprintf(stderr, "Invalid argument at \"-mf\".\n"); return 1; } }
		else if(!strcmp(o, "-ml")) { const int v = (int) need_int(argc, argv, &a, o); trim.min_len = v; par.minlen = v; cp.minlen = v; }       /* kma.c:581: one variable */
		else if(!strcmp(o, "-xl")) trim.max_len = (int) need_int(argc, argv, &a, o);
		else if(!strcmp(o, "-mp")) trim.min_phred = (int) need_int(argc, argv, &a, o);
		else if(!strcmp(o, "-mi")) trim.hardmask_q = (int) need_int(argc, argv, &a, o);
		else if(!strcmp(o, "-eq")) trim.min_q = (int) need_int(argc, argv, &a, o);
		else if(!strcmp(o, "-mq")) par.mq = (int) need_int(argc, argv, &a, o);
		else if(!strcmp(o, "-ts")) { par.ts = (int) need_int(argc, argv, &a, o); if(par.ts < 0 || par.ts > 30) { fprintf(stderr, "# Invalid seed trim parsed\n"); return 1; } }   /* kma.c:568-576 */
		else if(!strcmp(o, "-mrs")) { par.scoreT = need_num(argc, argv, &a, o); cp.mrs = par.scoreT; }
		else if(!strcmp(o, "-mrc")) par.mrc = need_num(argc, argv, &a, o);
		else if(!strcmp(o, "-mct")) cp.coverT = need_num(argc, argv, &a, o);
		else if(!strcmp(o, "-e") || !strcmp(o, "-p")) { evalue = need_num(argc, argv, &a, o); if(evalue < 0 || 1.0 < evalue) { fprintf(stderr, "Invalid argument at \"%s\".\n", o); return 1; } }
		else if(!strcmp(o, "-bcd")) bcd = (int) need_int(argc, argv, &a, o);
		else if(!strcmp(o, "-ID")) ID_t = need_num(argc, argv, &a, o);
		else if(!strcmp(o, "-md")) Depth_t = need_num(argc, argv, &a, o);
		else if(!strcmp(o, "-ex_mode")) par.exhaustive = 1;
		/* switches of the reference that change nothing in its result files: how the index is held (-mmap / -swap, kma.c:526), where its
		 * temporary files go (-tmp [dir/], kma.c:1031-1050: this program writes none), what it says on stderr (-status, -verbose [n]) */
		else if(!strcmp(o, "-mmap") || !strcmp(o, "-swap") || !strcmp(o, "-status")) { }
		else if(!strcmp(o, "-tmp") || !strcmp(o, "-verbose")) { if(a + 1 < argc && argv[a + 1][0] != '-') ++a; }
		else if(!strcmp(o, "-mem_mode")) mem_mode = 1;                                           /* kma.c:547 */
		else if(!strcmp(o, "-lc")) lc = 1;                                                      /* kma.c:694-701 */
		else if(!strcmp(o, "-and")) cmp_mode = 1;                                               /* kma.c:915-920 */
		else if(!strcmp(o, "-oa")) { cmp_mode = 2; ID_t = 1e-300; Depth_t = 0.0; }               /* (ID_t = 0 there; a row needs 0 < id anyway, and 0 means "the default" to kmahip_shard_opts) */
		else if(!strcmp(o, "-5p") || !strcmp(o, "-3p")) (void) need_int(argc, argv, &a, o);     /* parsed and handed to run_input*, where nothing reads them (runinput.c:127-368: phredStat never touches fiveClip / threeClip) */
		/* the scoring scheme (kma.c:821-915, 1024-1030): signs are forced like the reference's; -penalty is read and then overwritten
		 * by the mean of -transition and -transversion below, as there (kma.c:1308) */
		else if(!strcmp(o, "-reward")) par.rw.M = abs((int) need_int(argc, argv, &a, o));
		else if(!strcmp(o, "-penalty")) (void) need_int(argc, argv, &a, o);
		else if(!strcmp(o, "-gapopen")) par.rw.W1 = -abs((int) need_int(argc, argv, &a, o));
		else if(!strcmp(o, "-gapextend")) par.rw.U = -abs((int) need_int(argc, argv, &a, o));
		else if(!strcmp(o, "-localopen")) par.rw.Wl = -abs((int) need_int(argc, argv, &a, o));
		else if(!strcmp(o, "-Npenalty")) par.rw.Mn = -abs((int) need_int(argc, argv, &a, o));
		else if(!strcmp(o, "-per")) par.rw.PE = abs((int) need_int(argc, argv, &a, o));
		else if(!strcmp(o, "-transition")) Ts = -abs((int) need_int(argc, argv, &a, o));
		else if(!strcmp(o, "-transversion")) Tv = -abs((int) need_int(argc, argv, &a, o));
		else if(!strcmp(o, "-cge")) { par.scoreT = 0.5; cp.mrs = 0.5; par.rw.M = 1; par.rw.W1 = -5; par.rw.U = -1; par.rw.PE = 17; }      /* (its MM = -3 does not survive kma.c:1308) */
		else if(!strcmp(o, "-gpus")) gpus = (int) need_int(argc, argv, &a, o);
		else { fprintf(stderr, "kmahip_map: option %s is not one this program implements\n", o); usage(); return 2; }
	}
	if(!prefix || !input || !out) { fprintf(stderr, "kmahip_map: -i (or -ipe / -int), -t_db and -o are required\n"); usage(); return 2; }
	if(kmahip_set_cmp(cmp_mode)) { fprintf(stderr, "kmahip_map: %s\n", kmahip_last_error()); return 1; }
	{	/* the substitution matrix (kma.c:1307-1328) */
		par.rw.MM = (Ts + Tv - 1) / 2;
		for(int i = 0; i < 4; ++i) {
			for(int j = 0; j < 4; ++j) par.rw.d[i][j] = Tv;
			par.rw.d[i][4] = par.rw.Mn;
			par.rw.d[i][i ^ 2] = Ts;
			par.rw.d[i][i] = par.rw.M;
		}
		for(int j = 0; j < 5; ++j) par.rw.d[4][j] = par.rw.Mn;
		par.rw.d[4][4] = 0;
	}
	if(ref_fsa == 1) base_call = base_call == 1 ? 4 : 3;      /* kma.c:1278-1284: refNanoCaller / refCaller */
	/* like the reference: without -1t1 (and without -Mt1) the template finder is save_kmers_chain, reads may map in pieces */
	if(!one2one && !mt1) chain = 1;
	{	/* (without -apm the reference pairs by union, kma.c:206; -apm sets both stages, -pm stage 2 and -fpm stage 3a alone) */
		int s2 = apm == 1 ? 0 : (apm == 3 ? 2 : 1), s3 = s2;          /* 0 p, 1 u, 2 f */
		if(pm) s2 = pm == 1 ? 0 : (pm == 3 ? 2 : 1);
		if(fpm) s3 = fpm == 1 ? 0 : (fpm == 3 ? 2 : 1);
		/* forced pairing: stage 2 is built (save_kmers_forcePair: kmahip_scan_pe with apm = 2, pinned on the reference's -s2 tap); the whole
		 * run is not -- alnFragsForcePE is not built, and under -mem_mode a forced couple whose templates lie on both strands is filed with a
		 * negative count (runkma.c:1124), which sends its reads through anker_rc in stage 3c */
		if(s2 == 2 || s3 == 2) { fprintf(stderr, "kmahip_map: forced pairing (-apm f / -pm f / -fpm f) is not built beyond stage 2\n"); return 2; }

		par.apm = s2 | ((s3 + 1) << 4);          /* kmahip_params.apm: bits 0-1 stage 2, bits 4-5 stage 3a + 1 */
	}
	/* paired input without -1t1: couples are paired as ever, a record that lost its mate goes to the chain finder (savekmers.c:196-200) */
	const int pe_chain = chain && input2;
	if(lc && chain) { fprintf(stderr, "kmahip_map: -lc needs -1t1 (the chain finder's length-corrected anchors are not built)\n"); return 2; }
	if(kmahip_set_conclave_lc(lc)) return 1;
	if(mem_mode && !mt1) {          /* (runKMA_Mt1 comes before runKMA_MEM, kma.c:1598-1623: -Mt1 leaves -mem_mode without effect) */
		if(kmahip_set_mem_mode(1)) return 1;
	}
	if(pe_chain) chain = 0;
	/* -mrc in the default mode: mrchain (kmeranker.c:57-81) only drops templates when q_len < mrc * (the chain's span on the read), which
	 * no mrc <= 1 can make true -- stage 2 is as without it, the coverage test of stages 3a / 3c (mrcheck) is the aligner's own */
	if((chain || pe_chain) && par.mrc > 1.0) { fprintf(stderr, "kmahip_map: -mrc above 1 needs -1t1 (the chain finder's mrchain filter is not built)\n"); return 2; }
	/* -Mt1 with paired input: the reader is the paired one, the run is the single-end -Mt1 run over records of their own */
	const char *reader2 = input2;
	const int rc_mates = mt1 && input2;
	if(mt1) input2 = NULL;
	if(threads) {
		char v[16];
		snprintf(v, sizeof v, "%d", threads);
		setenv("KMAHIP_INGEST_THREADS", v, 1); setenv("KMAHIP_IO_THREADS", v, 1);          /* host threads of stage 1 and of the writers */
	}

	/* ranks */
	if(gpus > 1 && !getenv("KMAHIP_RANK")) return launch_ranks(gpus, argv);
	/* KMAHIP_MAP_EARLY_RETURN=1 (opt-in; never with a tool library preloaded, which has initialised the GPU already): the work is done
	 * by a child; when every output is closed it says so through a pipe and this process ends with its status, while the child gives
	 * back what it holds (gigabytes of mapped input, host arrays, the device context: 0.2-0.3 s of kernel work after a 10 M-read run)
	 * with nobody waiting. A wall time taken that way leaves the teardown out -- the reference's includes its own -- so it is not
	 * the default, and bench.py reports the one-process wall as the figure to compare. */
	int done_fd = -1;
	const char *early = getenv("KMAHIP_MAP_EARLY_RETURN");
	if(early && *early == '1' && !tool_preloaded() && !getenv("KMAHIP_MAP_NO_FORK") && !getenv("KMAHIP_RANK") && !getenv("RANK")) {
		int pfd[2];
		if(pipe(pfd) == 0) {
			const pid_t child = fork();
			if(child > 0) {
				unsigned char st = 0;
				close(pfd[1]);
				const ssize_t got = read(pfd[0], &st, 1);
				if(got == 1) _exit(st);                       /* outputs complete (or the child reported its failure) */
				int ws_ = 0;                                   /* the child ended without a word: its exit status tells */
				if(waitpid(child, &ws_, 0) < 0 || !WIFEXITED(ws_)) _exit(1);
				_exit(WEXITSTATUS(ws_));
			}
			if(child == 0) { close(pfd[0]); done_fd = pfd[1]; }
			else { close(pfd[0]); close(pfd[1]); }               /* fork failed: carry on in this process */
		}
	}
	g_done_fd = done_fd;
	int rank = 0, world = 1, local = 0;
	const char *key = getenv("KMAHIP_KEY");
	char keybuf[64];
	if(getenv("KMAHIP_RANK") && getenv("KMAHIP_WORLD")) { rank = atoi(getenv("KMAHIP_RANK")); world = atoi(getenv("KMAHIP_WORLD")); local = rank; }
	else if(getenv("RANK") && getenv("WORLD_SIZE")) {
		rank = atoi(getenv("RANK")); world = atoi(getenv("WORLD_SIZE")); local = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : rank;
		if(!key) { snprintf(keybuf, sizeof keybuf, "port%s", getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0"); key = keybuf; }
	}
	if(getenv("KMAHIP_SHARE_GPU")) local = 0;

	/* KMAHIP_COMM_FORCE_RCCL=1 on one rank: the sharded run over a real one-rank RCCL communicator (kmahip.h, kmahip_comm_describe) --
	 * every exchange of the N-rank run goes through RCCL, on a box with a single device */
	const int force_comm = world == 1 && getenv("KMAHIP_COMM_FORCE_RCCL") && getenv("KMAHIP_COMM_FORCE_RCCL")[0] == '1';
	const double t_start = now_s(), t_before_main = since_process_start();
	if(world == 1 && !force_comm && !getenv("KMAHIP_MAP_ONE_BATCH")) {
		/* the run batch by batch (kmahip_session_*: -1t1 single end and paired, the default mode, -Mt1): stage 1 of the next batch beside the
		 * device's work on this one, the host holding one batch at a time. A batch: a million records or a quarter of a gigabase,
		 * whichever comes first */
		stream_job sj;
		memset(&sj, 0, sizeof sj);
		sj.in1 = input; sj.in2 = reader2; sj.rc_mates = rc_mates; sj.trim = trim;
		sj.list1 = list1; sj.list2 = list2; sj.n_files = n_files;
		sj.batch_reads = getenv("KMAHIP_MAP_BATCH") ? atoll(getenv("KMAHIP_MAP_BATCH")) : 1000000;
		if(sj.batch_reads < 1) sj.batch_reads = 1;
		sj.batch_bases = getenv("KMAHIP_MAP_BATCH_BASES") ? atoll(getenv("KMAHIP_MAP_BATCH_BASES")) : (256ll << 20);
		pthread_mutex_init(&sj.mu, NULL); pthread_cond_init(&sj.cv, NULL);
		pthread_t reader;
		if(pthread_create(&reader, NULL, stream_main, &sj)) fail("cannot start a thread");
		kmahip_db *db; kmahip_ws *ws;
		if(kmahip_init(local) || kmahip_db_open(prefix, &db) || kmahip_ws_create(db, &ws)) die("open");
		if(pe_chain && kmahip_ws_set_pe_chain(ws, &cp)) die("open");
		const double t_open = now_s();
		kmahip_shard_opts so;
		memset(&so, 0, sizeof so);
		so.evalue = evalue; so.bcd = bcd; so.caller = base_call | (dense ? 16 : 0); so.sig90 = sig_mode; so.support = support; so.ref_fsa = ref_fsa; so.write_aln = !no_aln; so.max_frag = max_frag; so.ID_t = ID_t; so.Depth_t = Depth_t;
		int64_t hint = 0;
		{	/* (a guess at the number of reads from the size of the input: it only sizes the first allocation) */
			struct stat sb;
			for(int f = 0; f < n_files; ++f) {
				const size_t il = strlen(list1[f]);
				if(stat(list1[f], &sb) == 0) hint += (int64_t) (sb.st_size / (il > 3 && !strcmp(list1[f] + il - 3, ".gz") ? 60 : 300));
			}
		}
		kmahip_session *ses;
		char mt1_frag[4096];
		snprintf(mt1_frag, sizeof mt1_frag, "%s.frag.gz", out);
		if(reader2) hint *= 2;
		if(kmahip_session_open(db, ws, &par, &so, hint, &ses) || (chain && kmahip_session_set_chain(ses, &cp)) || (mt1 && kmahip_session_set_mt1(ses, mt1, one2one, no_frag ? NULL : mt1_frag)) ||
		   (input2 && kmahip_session_set_pe(ses))) die("session");
		int batches = 0;
		kmahip_db_info sinfo;
		int64_t unpinned = 0;
		int32_t longest = 0;
		if(kmahip_db_get_info(db, &sinfo)) die("open");
		for(;;) {
			pthread_mutex_lock(&sj.mu);
			while(sj.state == 0) pthread_cond_wait(&sj.cv, &sj.mu);
			const int st = sj.state;
			pthread_mutex_unlock(&sj.mu);
			if(st == 2) break;
			if(chain) {	/* (kmahip.h: the one place where the reference reads memory it never cleared) */
				int64_t c = 0;
				if(!kmahip_chain_unpinned_reads(sj.b.reads.len, sj.b.reads.N, sj.b.reads.N_off, sj.b.reads.n_reads, (int) sinfo.kmersize, longest, &longest, &c)) unpinned += c;
			}
			if(kmahip_session_upload(ses, &sj.b)) die("upload");
			pthread_mutex_lock(&sj.mu);
			sj.state = 0;
			pthread_cond_broadcast(&sj.cv);
			pthread_mutex_unlock(&sj.mu);
			if(kmahip_session_map(ses)) die("stages 2 + 3a");
			++batches;
		}
		pthread_join(reader, NULL);
		if(sj.rc) { fprintf(stderr, "kmahip_map: ingest: %s\n", sj.err); finish(1); }
		const double t_mapped = now_s();
		double ms[8];
		int64_t n_reads = 0, n_rows = 0;
		if(kmahip_session_finish(ses, out, !no_cons, !no_frag, &n_reads, &n_rows, ms)) die("finish");
		if(unpinned) fprintf(stderr, "# kmahip_map: %lld reads carry an N among their first k - 1 bases behind a longer read: the reference's records for them depend on what its buffer held\n", (long long) unpinned);
		fprintf(stderr, "# kmahip_map: %lld reads in %d batches, %lld fragment rows; wall: open %.2f s, ingest done after %.2f, mapped after %.2f, finish %.2f | uploads %.1f ms, stages 2+3a %.1f, "
		        "ConClave %.1f, traceback %.1f, pile-up + consensus %.1f, .res + .fsa %.1f, .frag.gz %.1f (+ %.1f beside the batches) (main entered %.2f s after process start; peak RSS %.0f MB)\n", (long long) n_reads, batches,
		        (long long) n_rows, t_open - t_start, sj.t_done - t_start, t_mapped - t_start, now_s() - t_mapped, ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], ms[6], ms[7], t_before_main, peak_rss_mb());
		if(getenv("KMAHIP_MAP_TEARDOWN")) {	/* what the process gives back, piece by piece and timed (the exit does the same in one go) */
			double t0 = now_s();
			kmahip_session_close(ses);
			const double t1 = now_s();
			kmahip_ws_destroy(ws);
			const double t2 = now_s();
			kmahip_db_close(db);
			const double t3 = now_s();
			kmahip_ingest_close(sj.ing);
			const double t4 = now_s();
			fprintf(stderr, "# kmahip_map: teardown: session %.3f s, workspace %.3f, index %.3f, reader %.3f\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3);
		}
		finish(0);
	}
	if(n_files > 1) { fprintf(stderr, "kmahip_map: several input files are read batch by batch on one rank (not with -gpus, KMAHIP_MAP_ONE_BATCH or KMAHIP_COMM_FORCE_RCCL)\n"); finish(2); }
	/* stage 1: this rank's part of the input as one batch (the arrays stay owned by the reader), while the device and the index come up */
	ingest_job job;
	memset(&job, 0, sizeof job);
	job.in1 = input; job.in2 = reader2; job.rc_mates = rc_mates; job.trim = trim; job.part = rank; job.parts = world;
	pthread_t ingest_thread;
	if(pthread_create(&ingest_thread, NULL, ingest_main, &job)) fail("cannot start a thread");

	kmahip_db *db; kmahip_ws *ws; kmahip_db_info info;
	if(kmahip_init(local) || kmahip_db_open(prefix, &db) || kmahip_ws_create(db, &ws) || kmahip_db_get_info(db, &info)) die("open");
	if(pe_chain && kmahip_ws_set_pe_chain(ws, &cp)) die("open");
	kmahip_comm *comm = NULL;
	if(world > 1 || force_comm) {
		const char *backend = getenv("KMAHIP_COMM") ? getenv("KMAHIP_COMM") : "rccl";
		if(kmahip_comm_init(rank, world, key ? key : "kmahip", backend, &comm)) die("communicator");
	}
	const int64_t D = info.DB_size;
	const double t_open = now_s();
	if(getenv("KMAHIP_MAP_STOP") && !strcmp(getenv("KMAHIP_MAP_STOP"), "open")) { fprintf(stderr, "# kmahip_map: stopped after open: %.3f s in main, entered %.2f s after process start\n", t_open - t_start, t_before_main); finish(0); }
	pthread_join(ingest_thread, NULL);
	if(job.rc) { fprintf(stderr, "kmahip_map: ingest: %s\n", job.err); finish(1); }
	kmahip_ingest *ing = job.ing;
	kmahip_read_batch b = job.b;
	const double t_ingest = now_s();

	if(world > 1 || force_comm) {
		/* an input the reader could not cut by bytes was delivered whole: this rank keeps its share of the records */
		if(job.whole_input) {
			const int64_t n_all = b.reads.n_reads;
			int64_t lo = n_all * rank / world, hi = n_all * (rank + 1) / world;
			if(lo < n_all && b.pair[lo] == 2) ++lo;          /* (a pair is never cut: the second mate goes with the first) */
			if(hi < n_all && b.pair[hi] == 2) ++hi;
			const int64_t so = b.reads.seq_off[lo], no = b.reads.N_off[lo], co = b.name_off[lo];
			int64_t *seq_off = xcalloc((size_t) (hi - lo) + 1, 8), *N_off = xcalloc((size_t) (hi - lo) + 1, 8), *name_off = xcalloc((size_t) (hi - lo) + 1, 8);
			for(int64_t i = lo; i <= hi; ++i) { seq_off[i - lo] = b.reads.seq_off[i] - so; N_off[i - lo] = b.reads.N_off[i] - no; name_off[i - lo] = b.name_off[i] - co; }
			b.reads.n_reads = hi - lo; b.reads.seq += so; b.reads.seq_off = seq_off; b.reads.len += lo; b.reads.N += no; b.reads.N_off = N_off;
			b.reads.seq_words = seq_off[hi - lo]; b.reads.N_total = N_off[hi - lo];
			b.names += co; b.name_off = name_off; b.pair += lo; b.records = hi - lo;
		}
		kmahip_shard_opts so;
		memset(&so, 0, sizeof so);
		so.evalue = evalue; so.bcd = bcd; so.caller = base_call | (dense ? 16 : 0); so.sig90 = sig_mode; so.support = support; so.ref_fsa = ref_fsa; so.write_aln = !no_aln; so.max_frag = max_frag; so.ID_t = ID_t; so.Depth_t = Depth_t;
		double ms[8];
		if(mt1 ? kmahip_run_mt1_sharded(db, ws, comm, &b, mt1, one2one, &par, &so, out, ms)
		       : chain ? kmahip_run_chain_sharded(db, ws, comm, &b, &par, &cp, &so, out, ms)
		       : (input2 ? kmahip_run_pe_sharded(db, ws, comm, &b, &par, &so, out, ms) : kmahip_run_se_sharded(db, ws, comm, &b, &par, &so, out, ms))) die("sharded run");
		if(rank == 0) {
			char path[4096];
			if(no_cons) { snprintf(path, sizeof path, "%s.fsa", out); remove(path); }
			if(no_frag) { snprintf(path, sizeof path, "%s.frag.gz", out); remove(path); }
			if(no_aln) { snprintf(path, sizeof path, "%s.aln", out); remove(path); }
		}
		fprintf(stderr, "# kmahip_map rank %d of %d: %lld reads; wall: ingest %.2f s beside open %.2f, run %.2f | upload %.1f ms, stages 2+3a %.1f, exchanges 1+2 + ConClave %.1f, "
		        "traceback %.1f, gather by owner %.1f, pile-up + consensus %.1f, writers %.1f, merge %.1f\n", rank, world, (long long) b.reads.n_reads, job.t_done - t_start,
		        t_open - t_start, now_s() - t_ingest, ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], ms[6], ms[7]);
		{	/* what carried the exchanges, as the transport itself reports it (a SCALE line can be checked for "RCCL saw N ranks") */
			char line[256];
			kmahip_comm_describe(comm, line, sizeof line);
			fprintf(stderr, "# kmahip_map rank %d comm: %s\n", rank, line);
		}
		finish(0);
	}
	const int64_t n = b.reads.n_reads;

	/* everything on the device, one call */
	int64_t tbases = 0;
	{	/* consensus capacity: template bases (from <prefix>.length.b: DB_size ints, the first is the k-mer index size) + slack */
		char path[4096];
		snprintf(path, sizeof path, "%s.length.b", prefix);
		FILE *f = fopen(path, "rb");
		int32_t v, i = 0;
		if(!f) { fprintf(stderr, "kmahip_map: cannot open %s\n", path); finish(1); }
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
	run.caller = base_call | (ref_fsa == 2 ? 8 : 0) | (dense ? 16 : 0) | (no_aln ? 0 : 32); run.sig90 = sig_mode; run.support = support;      /* -bcNano, -bc90, -bc, -bcg, -ref_fsa (bit 3: mark the trimmed insertion columns) */
	touch_job tj = { { (char *) run.tmpl, (char *) run.n_hits, (char *) run.rc, (char *) run.trace_stats },
	                 { ((size_t) n + 1) * 4, ((size_t) n + 1) * 4, ((size_t) n + 1) * 4, ((size_t) n * 10 + 10) * 4 } };
	pthread_t touch_thread;
	const int touching = !input2 && !chain && n > 100000 && !getenv("KMAHIP_MAP_NO_TOUCH") && !pthread_create(&touch_thread, NULL, touch_main, &tj);
	char fpath[4096];
	snprintf(fpath, sizeof fpath, "%s.frag.gz", out);
	const char *dev_frag = no_frag ? NULL : fpath;        /* (the paired and the default-mode run write the fragment file themselves) */
	if(mt1) {
		kmahip_assemble_opts ao;
		memset(&ao, 0, sizeof ao);
		ao.evalue = evalue; ao.bcd = bcd; ao.order = 1; ao.caller = base_call | (ref_fsa == 2 ? 8 : 0) | (dense ? 16 : 0) | (no_aln ? 0 : 32); ao.sig90 = sig_mode; ao.support = support;
		if(kmahip_run_mt1(db, ws, &b.reads, mt1, one2one, &par, &ao, &run)) die("kmahip_run_mt1");
	} else if(chain) {
		if(kmahip_run_chain(db, ws, &b.reads, b.names, b.name_off, &par, &cp, evalue, bcd, max_frag, dev_frag, &run)) die("kmahip_run_chain");
		{	/* (kmahip.h: the one place where the reference reads memory it never cleared) */
			int64_t unpinned = 0;
			if(!kmahip_chain_unpinned_reads(b.reads.len, b.reads.N, b.reads.N_off, b.reads.n_reads, (int) info.kmersize, 0, NULL, &unpinned) && unpinned)
				fprintf(stderr, "# kmahip_map: %lld reads carry an N among their first k - 1 bases behind a longer read: the reference's records for them depend on what its buffer held\n", (long long) unpinned);
		}
	} else if(input2) { if(kmahip_run_pe(db, ws, &b, &par, evalue, bcd, max_frag, dev_frag, &run)) die("kmahip_run_pe"); }
	else if(kmahip_run_se(db, ws, &b.reads, &par, evalue, bcd, max_frag, &run)) die("kmahip_run_se");

	if(touching) pthread_join(touch_thread, NULL);
	const double t_run = now_s();
	if(getenv("KMAHIP_MAP_STOP") && !strcmp(getenv("KMAHIP_MAP_STOP"), "run")) { fprintf(stderr, "# kmahip_map: stopped after the device run: %.3f s in main\n", t_run - t_start); finish(0); }
	/* out.res + out.fsa: names from <prefix>.name, one per line, in template order */
	char path[4096], *name = xcalloc(1 << 16, 1), *line = xcalloc((1 << 16) + 512, 1);
	snprintf(path, sizeof path, "%s.name", prefix);
	FILE *names = fopen(path, "r");
	snprintf(path, sizeof path, "%s.res", out);
	FILE *res = fopen(path, "w");
	snprintf(path, sizeof path, "%s.fsa", out);
	FILE *fsa = no_cons ? NULL : fopen(path, "w");
	snprintf(path, sizeof path, "%s.aln", out);
	FILE *aln = no_aln ? NULL : fopen(path, "w");
	if(!names || !res || (!fsa && !no_cons) || (!aln && !no_aln)) fail("cannot open the name file or the outputs");
	fputs("#Template\tScore\tExpected\tTemplate_length\tTemplate_Identity\tTemplate_Coverage\tQuery_Identity\tQuery_Coverage\tDepth\tq_value\tp_value\n", res);
	int64_t r = 0;
	size_t fsa_cap = 1 << 16;
	char *fsa_buf = xcalloc(fsa_cap, 1);
	for(int64_t t = 1; t < D && fgets(name, 1 << 16, names); ++t) {
		name[strcspn(name, "\n")] = 0;
		while(r < run.n_rows && run.rows[r].template_id < t) ++r;
		if(!(r < run.n_rows && run.rows[r].template_id == t && run.rows[r].significant)) continue;
		if(!kmahip_res_line(name, &run.rows[r], run.assembly.cover[t], run.assembly.aln_len[t], run.assembly.depth[t], ID_t, Depth_t, line, (1 << 16) + 512)) continue;
		fputs(line, res);
		const char *c = run.assembly.consensus + run.assembly.consensus_off[t];
		const size_t clen = strlen(c);
		if((clen / 60 + 2) * 224 + strlen(name) + 16 > fsa_cap) { fsa_cap = 2 * ((clen / 60 + 2) * 224 + strlen(name) + 16); free(fsa_buf); fsa_buf = xcalloc(fsa_cap, 1); }
		if(aln) {	/* printConsensus (printconsensus.c:26-37): the template's block of the alignment file */
			const int64_t got = kmahip_aln_entry(db, (int32_t) t, name, c, fsa_buf, (int64_t) fsa_cap);
			if(got < 0) die("kmahip_aln_entry");
			fwrite(fsa_buf, 1, (size_t) got, aln);
		}
		if(!fsa) continue;
		/* printConsensus (printconsensus.c:38-60): the consensus line without its '-' columns, 60 per line (an insertion column carries
		 * bit 7 when the .aln file was asked for; a gap called there was trimmed from the alignment, assembly.c:2094-2119) */
		fprintf(fsa, ">%s\n", name);
		char *o = fsa_buf;
		int col = 0;
		for(; *c; ++c) {
			const char b = (char) (*c & 0x7F);
			if(b == '_' || (b == '-' && (ref_fsa != 2 || (*c & 0x80)))) continue;
			*o++ = b;
			if(++col == 60) { *o++ = '\n'; col = 0; }
		}
		if(col) *o++ = '\n';
		fwrite(fsa_buf, 1, (size_t) (o - fsa_buf), fsa);
	}
	fclose(names); fclose(res);
	if(fsa) fclose(fsa);
	if(aln) fclose(aln);

	const double t_res = now_s();
	/* out.frag.gz (the paired and the default-mode run have written it themselves: their fragments are in record order, not read order) */
	int64_t frag_rows = 0;
	if(!no_frag && !input2 && !chain && kmahip_frag_write2(fpath, db, &b.reads, run.rc, run.tmpl, run.n_hits, run.trace_stats, max_frag, mt1 ? 1 : 0, b.names, b.name_off, &frag_rows)) die("kmahip_frag_write");
	const double t_frag = now_s();
	fprintf(stderr, "# kmahip_map: %lld reads, %lld fragment rows; wall: ingest %.2f s beside open %.2f (both done after %.2f), device run %.2f, .res + .fsa %.2f, .frag.gz %.2f | "
	        "upload %.1f ms, stages 2+3a %.1f, ConClave %.1f, traceback %.1f, pile-up + consensus %.1f, columns back %.1f%s (main entered %.2f s after process start)\n", (long long) n, (long long) frag_rows,
	        job.t_done - t_start, t_open - t_start, t_ingest - t_start, t_run - t_ingest, t_res - t_run, t_frag - t_res, run.ms[0], run.ms[1], run.ms[2], run.ms[3], run.ms[4],
	        input2 || chain ? 0.0 : run.ms[5],
	        input2 || chain ? " (the device run wrote the .frag.gz)" : "", t_before_main);
	/* every output is closed; the process ends here instead of unmapping gigabytes of reads and scratch one by one
	 * (KMAHIP_MAP_TEARDOWN=1: release everything in order, e.g. under a leak checker) */
	if(getenv("KMAHIP_MAP_TEARDOWN")) {
		kmahip_ws_destroy(ws);
		kmahip_db_close(db);
		kmahip_ingest_close(ing);
		finish(0);
	}
	finish(0);
}
