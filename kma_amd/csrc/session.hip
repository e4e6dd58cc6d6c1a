// session.hip -- the single-end `-1t1` run fed batch by batch (kmahip.h: kmahip_session_*): what runKMA does between its input stream
// and its output files (runkma.c:104-900) for an input that arrives in pieces, with the host holding one piece at a time.
//
// The reference streams: stage 1 writes records into a pipe, stage 2 and 3a take them as they come and spill frag_raw records to a
// temporary file (updatescores.c:283-295, tmp.c:27), ConClave and the assembly read that file back once the input has ended
// (runkma.c:563-594, 757-863). Here HBM is the temporary file:
//   add     a batch of stage-1 records goes up ONCE -- packed reads, N positions, headers -- behind what is there already (arrays
//           that grow by doubling), stages 2 and 3a run on it and add into the two ConClave vectors; the host's copy can go;
//   finish  ConClave per batch on the finished vectors, the `.res` statistics, the traceback per batch, one pile-up + consensus
//           over everything, and the fragment rows: ordered, measured and FORMATTED on the device (the reads and headers are
//           there), brought back as text a chunk at a time and compressed by the host's threads while the next chunk is made.
// Host memory is bounded by a batch and a few text chunks whatever the input's size; stage 1 of the next batch runs beside the
// device work of this one when the caller reads ahead on a thread of its own (examples/kmahip_map.c).
#include "pipeline_util.h"
#include <atomic>
#include <thread>
#include <string>

struct kmahip_gzstream;
kmahip_gzstream *kmahip_gzstream_open(const char *path);                                                        // fragout.hip
void kmahip_gzstream_submit(kmahip_gzstream *g, const char *text, size_t bytes, std::atomic<int> *done);
int kmahip_gzstream_close(kmahip_gzstream *g);
int kmahip_write_res_fsa(kmahip_db *db, const char *res_path, const char *fsa_path, bool header, const kmahip_res_row *rows, int64_t n_rows,
                         const int32_t *owner, int rank, const int64_t *cover, const int64_t *aln_len, const int64_t *depth, const char *cons,
                         const int64_t *cons_off, double ID_t, double Depth_t, int ref_fsa, const char *aln_path);                                    // pipeline.hip

namespace {

// a device array that grows by doubling (what it holds is copied over)
struct DevArr {
	char *p = nullptr;
	size_t cap = 0;
	~DevArr() { if(p) (void) hipFree(p); }
	int ensure(size_t need, size_t used, hipStream_t s) {
		if(need <= cap) return KMAHIP_OK;
		const size_t want = std::max(need, cap * 2);
		char *q = nullptr;
		if(hipMalloc((void **) &q, want) != hipSuccess && (want == need || hipMalloc((void **) &q, need) != hipSuccess)) { kmahip_set_error("hipMalloc of %zu bytes failed", need); return KMAHIP_ENOMEM; }
		if(used && hipMemcpyAsync(q, p, used, hipMemcpyDeviceToDevice, s) != hipSuccess) { (void) hipFree(q); kmahip_set_error("device copy failed"); return KMAHIP_EDEVICE; }
		if(hipStreamSynchronize(s) != hipSuccess) { (void) hipFree(q); kmahip_set_error("device copy failed"); return KMAHIP_EDEVICE; }
		if(p) (void) hipFree(p);
		p = q; cap = want;
		return KMAHIP_OK;
	}
	template <class T> T *as() const { return (T *) p; }
};

struct Batch {
	int64_t r0 = 0, n = 0, total = 0;
	int max_len = 0;
	kmahip_cands c{};
	kmahip_hits h{};
	std::vector<void *> owned;
	// the default mode: the batch's READS wait here (their own arrays) until kmahip_session_map has made the records out of them
	kmahip_reads tmp{};
	std::vector<void *> tmp_owned;
	int64_t read0 = 0;             // first read of the batch among all reads (its headers' index)
	int64_t words = 0;             // packed words of the batch's reads
	KmaChainRecs cr;               // the records' template lists live in its block
	void release_tmp() { for(void *q : tmp_owned) (void) hipFree(q); tmp_owned.clear(); }
	void release() { for(void *q : owned) (void) hipFree(q); owned.clear(); release_tmp(); kmahip_chain_records_free(&cr); }
};

template <class T> int dev_new(std::vector<void *> &owned, size_t n, T **out, bool zero, hipStream_t s) {
	void *d = nullptr;
	const size_t bytes = (n ? n : 1) * sizeof(T);
	if(hipMalloc(&d, bytes) != hipSuccess) { kmahip_set_error("hipMalloc of %zu bytes failed", bytes); return KMAHIP_ENOMEM; }
	owned.push_back(d);
	if(zero && hipMemsetAsync(d, 0, bytes, s) != hipSuccess) { kmahip_set_error("hipMemset failed"); return KMAHIP_EDEVICE; }
	*out = (T *) d;
	return KMAHIP_OK;
}

__global__ __launch_bounds__(256) void add_u64_kernel(int64_t n, const unsigned long long *a, unsigned long long *b) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i < n && a[i]) b[i] += a[i];
}
__global__ __launch_bounds__(256) void add_off_kernel(int64_t n, int64_t *v, int64_t base) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i < n) v[i] += base;
}

// ---- the fragment rows on the device (updateFrags, assembly.c:49-83; the host form is fragout.hip) -------------------------------
// filed: ConClave gave the read a template; kept: it also passed the stage-3c filter and gets a row
__global__ __launch_bounds__(256) void row_flags_kernel(int64_t n, const int32_t *tmpl, const int32_t *stats, int64_t *filed, int64_t *kept) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i > n) return;
	const bool f = i < n && tmpl[i] != 0;
	filed[i] = f; kept[i] = f && stats[10 * i + 3] != 0;
}
// the order assemble_KMA meets the fragments in: templates ascending; inside a template the chunks of max_frag filed fragments in
// stream order, each chunk back to front (conclave.c:164-166, 194)
__global__ __launch_bounds__(256) void row_keys_kernel(int64_t n, const int32_t *tmpl, const int64_t *kept, const int64_t *rank, int64_t max_frag, int order, unsigned long long *keys, int64_t *vals) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n) return;
	vals[i] = i;
	if(!(kept[i + 1] - kept[i])) { keys[i] = ~0ull; return; }
	if(order == 1) { keys[i] = ((unsigned long long) abs(tmpl[i]) << 40) | (unsigned long long) rank[i]; return; }          // (`-Mt1`: as the stream has them)
	const unsigned long long rk = (unsigned long long) rank[i], chunk = rk / (unsigned long long) max_frag, in = rk % (unsigned long long) max_frag;
	keys[i] = ((unsigned long long) abs(tmpl[i]) << 40) | (chunk * (unsigned long long) max_frag + ((unsigned long long) max_frag - 1ull - in));
}

struct RowArgs {
	const uint64_t *seq;
	const int64_t *seq_off, *N_off, *name_off;
	const int32_t *len, *N, *rc, *tmpl, *n_hits, *stats;
	const char *names, *tnames;
	const int64_t *tname_off;
	const int64_t *row_read;
	const int64_t *name_idx;     // the header a row carries (NULL: the read's own; the default mode: the read its record came from)
	int64_t *row_off;
};
__device__ __forceinline__ int digits_of(int v) {
	unsigned u = v < 0 ? 0u - (unsigned) v : (unsigned) v;
	int d = v < 0 ? 2 : 1;
	while(u >= 10u) { u /= 10u; ++d; }
	return d;
}
__global__ __launch_bounds__(256) void row_len_kernel(const RowArgs A, int64_t n_rows) {
	const int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(r > n_rows) return;
	if(r == n_rows) { A.row_off[r] = 0; return; }
	const int64_t i = A.row_read[r];
	const int t = abs(A.tmpl[i]);
	const int32_t *st = A.stats + 10 * i;
	// bases, "\t<ties>\t<score>\t<start>\t<end>\t<template>\t<header>\n"
	A.row_off[r] = (int64_t) A.len[i] + 4 + digits_of(A.n_hits[i]) + digits_of(st[0]) + digits_of(st[1]) + digits_of(st[2]) + 1 + (A.tname_off[t] - A.tname_off[t - 1]) + 1 +
	               (A.name_off[(A.name_idx ? A.name_idx[i] : i) + 1] - A.name_off[A.name_idx ? A.name_idx[i] : i] - 1) + 1;
}
__device__ __forceinline__ char *put_int_dev(char *o, int v) {
	*o++ = '\t';
	unsigned u = v < 0 ? 0u - (unsigned) v : (unsigned) v;
	if(v < 0) *o++ = '-';
	char d[12];
	int k = 0;
	do { d[k++] = (char) ('0' + u % 10u); u /= 10u; } while(u);
	while(k) *o++ = d[--k];
	return o;
}
// one thread per row: the read as it was aligned (reverse complemented when it was filed on the minus strand), four bases per
// 32-bit store; N's; the figures; the two names
__global__ __launch_bounds__(256) void row_format_kernel(const RowArgs A, int64_t r0, int64_t r1, int64_t text_base, char *text) {
	const int64_t r = r0 + (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(r >= r1) return;
	const int64_t i = A.row_read[r];
	char *o = text + (A.row_off[r] - text_base);
	const int L = A.len[i];
	const uint64_t *w = A.seq + A.seq_off[i];
	const int tt = A.tmpl[i];
	const bool flip = ((A.rc[i] & 1) != 0) != (tt < 0);
	const uint32_t lut = 0x54474341u;          // 'A' 'C' 'G' 'T', low byte first
	for(int b = 0; b < L; b += 4) {
		uint32_t four = 0;
#pragma unroll
		for(int x = 0; x < 4; ++x) {
			const int pos = b + x;
			int code = 0;
			if(pos < L) {
				const int src = flip ? L - 1 - pos : pos;
				code = (int) ((w[src >> 5] >> (62 - ((src & 31) << 1))) & 3ull);
				if(flip) code = 3 - code;
			}
			four |= ((lut >> (8 * code)) & 0xFFu) << (8 * x);
		}
		if(b + 4 <= L) memcpy(o + b, &four, 4);          // (unaligned: rows begin anywhere)
		else for(int x = 0; b + x < L; ++x) o[b + x] = (char) ((four >> (8 * x)) & 0xFFu);
	}
	const int32_t *Np = A.N + A.N_off[i];
	const int nN = (int) (A.N_off[i + 1] - A.N_off[i]);
	for(int x = 0; x < nN; ++x) o[flip ? L - 1 - Np[x] : Np[x]] = 'N';
	o += L;
	const int32_t *st = A.stats + 10 * i;
	o = put_int_dev(o, A.n_hits[i]); o = put_int_dev(o, st[0]); o = put_int_dev(o, st[1]); o = put_int_dev(o, st[2]);
	*o++ = '\t';
	const int t = abs(tt);
	for(int64_t x = A.tname_off[t - 1]; x < A.tname_off[t]; ++x) *o++ = A.tnames[x];
	*o++ = '\t';
	const int64_t ni = A.name_idx ? A.name_idx[i] : i;
	for(int64_t x = A.name_off[ni]; x < A.name_off[ni + 1] - 1; ++x) *o++ = A.names[x];
	*o++ = '\n';
}
__global__ __launch_bounds__(256) void row_blocks_kernel(int64_t n_blocks, int64_t rows_per_block, int64_t n_rows, const int64_t *row_off, int64_t *block_off) {
	const int64_t b = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(b <= n_blocks) block_off[b] = row_off[b * rows_per_block < n_rows ? b * rows_per_block : n_rows];
}

// `-Mt1`: a read is kept when its traceback gave an alignment; the `.res` row's Score = the sum of KMA()'s own scores of the kept reads,
// without the end bonus the read filter added (alnToMat, assembly.c:1328-1334; kmahip_run_mt1)
__global__ __launch_bounds__(256) void session_mt1_kept_kernel(int64_t n, const int32_t *stats, int32_t tmpl, int t_len, int Wl, int32_t *o_tmpl, int32_t *o_nh, unsigned long long *sum) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n) return;
	const int32_t *st = stats + 10 * i;
	const bool kept = st[3] != 0;
	o_tmpl[i] = kept ? tmpl : 0;
	o_nh[i] = 1;
	if(kept) atomicAdd(sum, (unsigned long long) (st[0] - Wl * ((st[1] == 0) + (st[2] == t_len))));
}

}  // namespace

struct kmahip_session {
	kmahip_db *db = nullptr;
	kmahip_ws *ws = nullptr;
	kmahip_params par{};
	kmahip_shard_opts opts{};
	DevArr seq, seq_off, len, N, N_off, names, name_off;
	// the default mode (kmahip_session_set_chain): the arrays above hold RECORDS, with their query bounds and the read they came from;
	// names / name_off stay per read
	bool chain = false;
	// paired input (kmahip_session_set_pe): the batches' reads and headers wait in HBM like the single-end ones, the mates' flags on the
	// host (a byte per read); the stages run when the input has ended -- ConClave reads the finished score vectors anyway, and the
	// stages before it take a tenth of a second per ten million pairs --, the first batch is run through stages 2 and 3a once and
	// thrown away while stage 1 reads on (first launches, scratch: what a separate warm-up run used to pay for)
	bool pe = false, pe_warm = false;
	std::vector<uint8_t> pair;
	kmahip_chain_params cp{};
	DevArr qs, qe, rread;
	int64_t n_reads = 0;
	// `-Mt1` (kmahip_session_set_mt1): the reads of every batch are traced as the batch comes; figures, strands and runs of all of them
	int32_t mt1 = 0;
	int mt1_one2one = 0;
	DevArr t_stats, t_off, t_nops, t_rc, t_pool, t_tmpl, t_nh;
	int64_t pool_used = 0;
	unsigned long long *mt1_sum = nullptr;          // Score of the `.res` row so far
	KmaFragSink *sink = nullptr;                    // the fragment file, written as the batches are traced (kmahip_session_set_mt1 with a path)
	int64_t sink_rows = 0;
	double ms_frag = 0;
	// the batch whose rows are still to be made: a thread makes them while the next batch is traced (or while everything is piled up)
	bool frag_have = false;
	int64_t frag_r0 = 0, frag_n = 0;
	int frag_max_len = 0, frag_rc = 0, device = 0;
	std::thread frag_thread;
	int64_t n = 0, words = 0, nN = 0, name_bytes = 0;
	int max_len = 0;
	uint64_t *AS = nullptr, *AS_batch = nullptr;          // 2 D each: alignment_scores | uniq_alignment_scores
	std::vector<Batch> batches, uploaded;          // mapped / uploaded and waiting for kmahip_session_map
	// the text chunks of the fragment rows: pinned host buffers, made when the session opens (pinning 64 MB takes ~10 ms: paid beside
	// stage 1 instead of in front of the writer)
	static constexpr int NBUF = 3;
	char *h_text[NBUF] = {nullptr, nullptr, nullptr};
	int64_t text_chunk = 0;
	double ms_upload = 0, ms_map = 0;
	~kmahip_session() {
		for(Batch &b : batches) b.release();
		for(Batch &b : uploaded) b.release();
		if(AS) (void) hipFree(AS);
		if(AS_batch) (void) hipFree(AS_batch);
		if(frag_thread.joinable()) frag_thread.join();
		if(mt1_sum) (void) hipFree(mt1_sum);
		if(sink) (void) kmahip_frag_sink_close(sink);          // (before the pinned buffers go)
		for(int x = 0; x < NBUF; ++x) if(h_text[x]) (void) hipHostFree(h_text[x]);
	}
};

extern "C" int kmahip_session_open(kmahip_db *db, kmahip_ws *ws, const kmahip_params *p, const kmahip_shard_opts *opts, int64_t reads_hint, kmahip_session **out) {
	if(!db || !ws || !p || !opts || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	kmahip_session *S = new kmahip_session();
	S->db = db; S->ws = ws; S->par = *p; S->opts = *opts;
	const size_t D = db->info.DB_size;
	if(hipMalloc((void **) &S->AS, 2 * D * 8) != hipSuccess || hipMalloc((void **) &S->AS_batch, 2 * D * 8) != hipSuccess || hipMemset(S->AS, 0, 2 * D * 8) != hipSuccess) {
		delete S; kmahip_set_error("hipMalloc failed"); return KMAHIP_ENOMEM;
	}
	// (room for the hinted number of reads of ~150 bases up front: no copies while the arrays grow)
	if(reads_hint > 0) {
		const size_t r = (size_t) reads_hint + (size_t) reads_hint / 16 + 1024;
		int rc;
		if((rc = S->seq.ensure(r * 6 * 8, 0, 0)) || (rc = S->seq_off.ensure((r + 1) * 8, 0, 0)) || (rc = S->len.ensure((r + 1) * 4, 0, 0)) || (rc = S->N_off.ensure((r + 1) * 8, 0, 0)) ||
		   (rc = S->name_off.ensure((r + 1) * 8, 0, 0)) || (rc = S->names.ensure(r * 12, 0, 0))) { delete S; return rc; }
	}
	S->text_chunk = getenv("KMAHIP_FRAG_CHUNK") ? std::max<int64_t>(1024, atoll(getenv("KMAHIP_FRAG_CHUNK"))) : (64ll << 20);
	for(int x = 0; x < kmahip_session::NBUF; ++x) if(hipHostMalloc((void **) &S->h_text[x], (size_t) S->text_chunk + 16, hipHostMallocDefault) != hipSuccess) { S->h_text[x] = nullptr; delete S; kmahip_set_error("hipHostMalloc failed"); return KMAHIP_ENOMEM; }
	*out = S;
	return KMAHIP_OK;
}

extern "C" void kmahip_session_close(kmahip_session *S) { delete S; }

extern "C" int kmahip_session_set_mt1(kmahip_session *S, int32_t tmpl, int one2one, const char *frag_path) {
	if(!S) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	if(S->n || S->n_reads || !S->uploaded.empty() || S->chain || S->mt1 || S->pe) { kmahip_set_error("the mode of a session is chosen before its first batch"); return KMAHIP_EINVAL; }
	if(tmpl < 1 || (size_t) tmpl >= S->db->info.DB_size) { kmahip_set_error("template %d out of range", tmpl); return KMAHIP_EINVAL; }
	if(hipMalloc((void **) &S->mt1_sum, 8) != hipSuccess || hipMemset(S->mt1_sum, 0, 8) != hipSuccess) { kmahip_set_error("hipMalloc failed"); return KMAHIP_ENOMEM; }
	if(frag_path) {
		int rc = kmahip_db_load_names(S->db);
		if(rc) return rc;
		if(!(S->sink = kmahip_frag_sink_open(frag_path))) return KMAHIP_EIO;
	}
	S->mt1 = tmpl; S->mt1_one2one = one2one;
	(void) hipGetDevice(&S->device);
	return KMAHIP_OK;
}

extern "C" int kmahip_session_set_pe(kmahip_session *S) {
	if(!S) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	if(S->n || S->n_reads || !S->uploaded.empty() || S->chain || S->mt1) { kmahip_set_error("the mode of a session is chosen before its first batch"); return KMAHIP_EINVAL; }
	S->pe = true;
	return KMAHIP_OK;
}

extern "C" int kmahip_session_set_chain(kmahip_session *S, const kmahip_chain_params *cp) {
	if(!S) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	if(S->n || S->n_reads || !S->uploaded.empty() || S->mt1 || S->pe) { kmahip_set_error("the mode of a session is chosen before its first batch"); return KMAHIP_EINVAL; }
	S->chain = true;
	if(cp) S->cp = *cp; else { S->cp.minlen = 16; S->cp.pad_ = 0; S->cp.coverT = 0.1; S->cp.mrs = 0.5; }
	return KMAHIP_OK;
}

// One batch of stage-1 records, first half: uploaded behind the batches before it. The caller's arrays are free when this returns
// (a reader thread may go on to the next batch while kmahip_session_map works on this one).
extern "C" int kmahip_session_upload(kmahip_session *S, const kmahip_read_batch *batch) {
	if(!S || !batch) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	const kmahip_reads &R = batch->reads;
	const int64_t nb = R.n_reads;
	if(nb < 0 || R.seq_words < 0 || R.N_total < 0) { kmahip_set_error("negative size"); return KMAHIP_EINVAL; }
	if(nb == 0) return KMAHIP_OK;
	if(!batch->names || !batch->name_off) { kmahip_set_error("the batch carries no read names"); return KMAHIP_EINVAL; }
	if(S->pe) {
		if(!batch->pair) { kmahip_set_error("a paired session needs the batch's mate flags"); return KMAHIP_EINVAL; }
		// (the reader never cuts a couple: a batch ends behind a second mate or a single read)
		if(batch->pair[nb - 1] == 1) { kmahip_set_error("a batch of a paired session ends inside a couple"); return KMAHIP_EINVAL; }
		S->pair.insert(S->pair.end(), batch->pair, batch->pair + nb);
	}
	hipStream_t s = 0;
	auto t = std::chrono::steady_clock::now();
	int rc;
	const int64_t nbytes = batch->name_off[nb];
	if(S->chain) {
		// the default mode: the headers behind those of the reads before, the reads into arrays of the batch's own (the records made of
		// them are what goes behind the records before: kmahip_session_map)
		if((rc = S->names.ensure((size_t) (S->name_bytes + nbytes + 1), (size_t) S->name_bytes, s)) ||
		   (rc = S->name_off.ensure((size_t) (S->n_reads + nb + 1) * 8, (size_t) (S->n_reads + 1) * 8, s))) return rc;
		if(nbytes) HIP_TRY(hipMemcpyAsync(S->names.as<char>() + S->name_bytes, batch->names, (size_t) nbytes, hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(S->name_off.as<int64_t>() + S->n_reads, batch->name_off, (size_t) (nb + 1) * 8, hipMemcpyHostToDevice, s));
		if(S->name_bytes) hipLaunchKernelGGL(add_off_kernel, dim3((unsigned) ((nb + 1 + 255) / 256)), dim3(256), 0, s, nb + 1, S->name_off.as<int64_t>() + S->n_reads, S->name_bytes);
		Batch U;
		uint64_t *d_seq = nullptr;
		int64_t *d_so = nullptr, *d_no = nullptr;
		int32_t *d_len = nullptr, *d_N = nullptr;
		if((rc = dev_new(U.tmp_owned, (size_t) R.seq_words + 2, &d_seq, false, s)) || (rc = dev_new(U.tmp_owned, (size_t) nb + 1, &d_so, false, s)) || (rc = dev_new(U.tmp_owned, (size_t) nb + 1, &d_len, false, s)) ||
		   (rc = dev_new(U.tmp_owned, (size_t) R.N_total + 1, &d_N, false, s)) || (rc = dev_new(U.tmp_owned, (size_t) nb + 1, &d_no, false, s))) { U.release(); return rc; }
		if(R.seq_words) HIP_TRY(hipMemcpyAsync(d_seq, R.seq, (size_t) R.seq_words * 8, hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemsetAsync(d_seq + R.seq_words, 0, 16, s));
		HIP_TRY(hipMemcpyAsync(d_so, R.seq_off, (size_t) (nb + 1) * 8, hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(d_len, R.len, (size_t) nb * 4, hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemsetAsync(d_len + nb, 0, 4, s));
		if(R.N_total) HIP_TRY(hipMemcpyAsync(d_N, R.N, (size_t) R.N_total * 4, hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(d_no, R.N_off, (size_t) (nb + 1) * 8, hipMemcpyHostToDevice, s));
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(s));
		U.tmp = R;
		U.tmp.seq = d_seq; U.tmp.seq_off = d_so; U.tmp.len = d_len; U.tmp.N = d_N; U.tmp.N_off = d_no; U.tmp.q_start = nullptr; U.tmp.q_end = nullptr;
		U.read0 = S->n_reads; U.max_len = R.max_len;
		S->uploaded.push_back(std::move(U));
		S->n_reads += nb; S->name_bytes += nbytes;
		S->max_len = std::max(S->max_len, R.max_len);
		S->ms_upload += since(t);
		return KMAHIP_OK;
	}
	if((rc = S->seq.ensure((size_t) (S->words + R.seq_words + 2) * 8, (size_t) S->words * 8, s)) || (rc = S->seq_off.ensure((size_t) (S->n + nb + 1) * 8, (size_t) (S->n + 1) * 8, s)) ||
	   (rc = S->len.ensure((size_t) (S->n + nb + 1) * 4, (size_t) S->n * 4, s)) || (rc = S->N.ensure((size_t) (S->nN + R.N_total + 1) * 4, (size_t) S->nN * 4, s)) ||
	   (rc = S->N_off.ensure((size_t) (S->n + nb + 1) * 8, (size_t) (S->n + 1) * 8, s)) || (rc = S->names.ensure((size_t) (S->name_bytes + nbytes + 1), (size_t) S->name_bytes, s)) ||
	   (rc = S->name_off.ensure((size_t) (S->n + nb + 1) * 8, (size_t) (S->n + 1) * 8, s))) return rc;
	if(R.seq_words) HIP_TRY(hipMemcpyAsync(S->seq.as<uint64_t>() + S->words, R.seq, (size_t) R.seq_words * 8, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemsetAsync(S->seq.as<uint64_t>() + S->words + R.seq_words, 0, 16, s));
	HIP_TRY(hipMemcpyAsync(S->len.as<int32_t>() + S->n, R.len, (size_t) nb * 4, hipMemcpyHostToDevice, s));
	if(R.N_total) HIP_TRY(hipMemcpyAsync(S->N.as<int32_t>() + S->nN, R.N, (size_t) R.N_total * 4, hipMemcpyHostToDevice, s));
	if(nbytes) HIP_TRY(hipMemcpyAsync(S->names.as<char>() + S->name_bytes, batch->names, (size_t) nbytes, hipMemcpyHostToDevice, s));
	// the three offset arrays go up as they are and are moved behind what is there on the device
	HIP_TRY(hipMemcpyAsync(S->seq_off.as<int64_t>() + S->n, R.seq_off, (size_t) (nb + 1) * 8, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(S->N_off.as<int64_t>() + S->n, R.N_off, (size_t) (nb + 1) * 8, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(S->name_off.as<int64_t>() + S->n, batch->name_off, (size_t) (nb + 1) * 8, hipMemcpyHostToDevice, s));
	const unsigned g1 = (unsigned) ((nb + 1 + 255) / 256);
	if(S->words) hipLaunchKernelGGL(add_off_kernel, dim3(g1), dim3(256), 0, s, nb + 1, S->seq_off.as<int64_t>() + S->n, S->words);
	if(S->nN) hipLaunchKernelGGL(add_off_kernel, dim3(g1), dim3(256), 0, s, nb + 1, S->N_off.as<int64_t>() + S->n, S->nN);
	if(S->name_bytes) hipLaunchKernelGGL(add_off_kernel, dim3(g1), dim3(256), 0, s, nb + 1, S->name_off.as<int64_t>() + S->n, S->name_bytes);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s));          // (the host arrays are the caller's again)
	S->ms_upload += since(t);
	Batch U;
	U.r0 = S->n; U.n = nb; U.max_len = R.max_len; U.words = R.seq_words;
	S->uploaded.push_back(std::move(U));
	S->n += nb; S->n_reads += nb; S->words += R.seq_words; S->nN += R.N_total; S->name_bytes += nbytes;
	S->max_len = std::max(S->max_len, R.max_len);
	return KMAHIP_OK;
}

static int session_map_one(kmahip_session *S, Batch &B);
static int session_warm_pe(kmahip_session *S, Batch &B);

// ... second half: stages 2 and 3a on every batch that has been uploaded and not mapped yet
extern "C" int kmahip_session_map(kmahip_session *S) {
	if(!S) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	while(!S->uploaded.empty()) {
		Batch B = std::move(S->uploaded.front());
		S->uploaded.erase(S->uploaded.begin());
		const int rc = S->pe ? session_warm_pe(S, B) : session_map_one(S, B);
		if(rc) return rc;
	}
	return KMAHIP_OK;
}

extern "C" int kmahip_session_add(kmahip_session *S, const kmahip_read_batch *batch) {
	const int rc = kmahip_session_upload(S, batch);
	return rc ? rc : kmahip_session_map(S);
}

// paired session: nothing is mapped before the input has ended (kmahip_session_finish), but the first batch goes through the whole
// paired run once, results thrown away, while stage 1 reads on: the first launch of every kernel and the workspace's scratch are
// paid for beside the I/O (what examples/kmahip_map's warm-up on made-up reads was for)
static int session_pe_view(kmahip_session *S, int64_t r0, int64_t n, kmahip_read_batch *hb) {
	memset(hb, 0, sizeof *hb);
	kmahip_reads &W = hb->reads;
	W.n_reads = n; W.seq = S->seq.as<uint64_t>(); W.seq_off = S->seq_off.as<int64_t>() + r0; W.len = S->len.as<int32_t>() + r0; W.N = S->N.as<int32_t>(); W.N_off = S->N_off.as<int64_t>() + r0;
	W.seq_words = S->words; W.N_total = S->nN; W.max_len = S->max_len;
	hb->pair = S->pair.data() + r0;
	return KMAHIP_OK;
}

static int session_warm_pe(kmahip_session *S, Batch &B) {
	if(S->pe_warm || getenv("KMAHIP_SESSION_NO_WARM")) return KMAHIP_OK;
	S->pe_warm = true;
	int64_t n = std::min<int64_t>(B.n, 262144);
	while(n > 0 && S->pair[(size_t) (B.r0 + n - 1)] == 1) --n;          // (not inside a couple)
	if(n < 2) return KMAHIP_OK;
	auto t = std::chrono::steady_clock::now();
	const size_t D = S->db->info.DB_size;
	kmahip_read_batch hb;
	session_pe_view(S, B.r0, n, &hb);
	KmaPeDev pd{S->names.as<char>(), S->name_off.as<int64_t>() + B.r0, S->h_text, S->text_chunk, nullptr};
	std::vector<kmahip_res_row> rows(D);
	std::vector<int64_t> a(4 * D, 0);
	kmahip_run run;
	memset(&run, 0, sizeof run);
	run.rows = rows.data(); run.rows_cap = (int64_t) D;
	run.assembly.cover = a.data(); run.assembly.aln_len = a.data() + D; run.assembly.depth = a.data() + 2 * D; run.assembly.asm_len = a.data() + 3 * D;
	run.caller = S->opts.caller; run.sig90 = S->opts.sig90; run.support = S->opts.support;
	const int rc = kmahip_run_pe_resident(S->db, S->ws, &hb, &pd, &S->par, S->opts.evalue, S->opts.bcd, S->opts.max_frag, nullptr, &run);
	S->ms_map += since(t);
	return rc;
}

// `-Mt1`, the fragment rows of one traced batch: one template, stream order -- they follow those of the batches before. Made by a thread
// of their own on the file's stream while the main thread traces the next batch (kmahip_session_map) or piles everything up
// (kmahip_session_finish): both only read what the rows are made of, and the arrays are not moved meanwhile.
static void mt1_frag_start(kmahip_session *S) {
	if(!S->frag_have || !S->sink) return;
	if(S->frag_thread.joinable()) S->frag_thread.join();          // (a call that ended early left it behind)
	S->frag_have = false;
	S->frag_rc = 0;
	S->frag_thread = std::thread([S]() {
		(void) hipSetDevice(S->device);
		auto tf = std::chrono::steady_clock::now();
		const int64_t r0 = S->frag_r0;
		kmahip_reads w{};
		w.n_reads = S->frag_n; w.seq = S->seq.as<uint64_t>(); w.seq_off = S->seq_off.as<int64_t>() + r0; w.len = S->len.as<int32_t>() + r0; w.N = S->N.as<int32_t>();
		w.N_off = S->N_off.as<int64_t>() + r0; w.seq_words = S->words; w.N_total = S->nN; w.max_len = S->frag_max_len;
		int64_t rows = 0;
		S->frag_rc = kmahip_frag_write_dev(S->db, &w, S->names.as<char>(), S->name_off.as<int64_t>() + r0, nullptr, S->t_rc.as<int32_t>() + r0, S->t_tmpl.as<int32_t>() + r0, S->t_nh.as<int32_t>() + r0,
		                                   S->t_stats.as<int32_t>() + 10 * r0, nullptr, S->opts.max_frag > 0 ? S->opts.max_frag : 1000000, "", S->text_chunk, S->h_text, &rows, 1, S->sink);
		S->sink_rows += rows;
		S->ms_frag += since(tf);
	});
}
static int mt1_frag_join(kmahip_session *S) {
	if(S->frag_thread.joinable()) S->frag_thread.join();
	const int rc = S->frag_rc;
	S->frag_rc = 0;
	return rc;
}

// `-Mt1`: seeds + traceback of the batch's reads against the one template, into the session's arrays behind the batches before
static int session_map_mt1(kmahip_session *S, Batch &B, const kmahip_reads &d) {
	hipStream_t s = 0;
	const int64_t nb = B.n, r0 = B.r0, n1 = r0 + nb;
	int rc;
	if((rc = S->t_stats.ensure((size_t) (10 * n1 + 10) * 4, (size_t) r0 * 40, s)) || (rc = S->t_off.ensure((size_t) (n1 + 1) * 8, (size_t) r0 * 8, s)) ||
	   (rc = S->t_nops.ensure((size_t) (n1 + 1) * 4, (size_t) r0 * 4, s)) || (rc = S->t_rc.ensure((size_t) (n1 + 1) * 4, (size_t) r0 * 4, s)) ||
	   (rc = S->t_tmpl.ensure((size_t) (n1 + 1) * 4, (size_t) r0 * 4, s)) || (rc = S->t_nh.ensure((size_t) (n1 + 1) * 4, (size_t) r0 * 4, s))) return rc;
	HIP_TRY(hipMemsetAsync(S->t_stats.as<int32_t>() + 10 * r0, 0, (size_t) nb * 40, s));
	HIP_TRY(hipMemsetAsync(S->t_off.as<int64_t>() + r0, 0, (size_t) nb * 8, s));
	HIP_TRY(hipMemsetAsync(S->t_nops.as<int32_t>() + r0, 0, (size_t) nb * 4, s));
	HIP_TRY(hipMemsetAsync(S->t_rc.as<int32_t>() + r0, 0, (size_t) nb * 4, s));
	mt1_frag_start(S);          // (the batch before: nothing it reads is moved or written from here on)
	// (the run pool: a guess from the batch's bases, redone with the count the kernels report if short)
	int64_t room = B.words * 32 / 3 + 8 * nb + (1 << 16);
	unsigned long long used = 0;
	for(int attempt = 0;; ++attempt) {
		if((rc = S->t_pool.ensure((size_t) (S->pool_used + room) * 4, (size_t) S->pool_used * 4, s))) return rc;
		kmahip_traces tr;
		tr.stats = S->t_stats.as<int32_t>() + 10 * r0; tr.ops_off = S->t_off.as<int64_t>() + r0; tr.n_ops = S->t_nops.as<int32_t>() + r0;
		tr.ops = S->t_pool.as<uint32_t>() + S->pool_used; tr.ops_cap = (int64_t) (S->t_pool.cap / 4) - S->pool_used;
		if((rc = kmahip_launch_longtrace(S->db, S->ws, &d, nullptr, S->mt1, nullptr, nullptr, S->mt1_one2one, &S->par, &tr, S->t_rc.as<int32_t>() + r0, s))) return rc;
		used = 0;
		const int st = ws_status(S->ws, &used);
		if(st == 2 || (int64_t) used > tr.ops_cap) {
			if(attempt >= 2) { kmahip_set_error("alignment run pool: %llu runs needed", used); return KMAHIP_EOVERFLOW; }
			room = (int64_t) used + (1 << 16);
			continue;
		}
		break;
	}
	// (the runs of a read are addressed from the start of the pool)
	if(S->pool_used) hipLaunchKernelGGL(add_off_kernel, dim3((unsigned) ((nb + 255) / 256)), dim3(256), 0, s, nb, S->t_off.as<int64_t>() + r0, S->pool_used);
	// the kept reads, their part of the `.res` row's Score
	hipLaunchKernelGGL(session_mt1_kept_kernel, dim3((unsigned) ((nb + 255) / 256)), dim3(256), 0, s, nb, S->t_stats.as<int32_t>() + 10 * r0, S->mt1, S->db->h_tlen[(size_t) S->mt1], S->par.rw.Wl,
	                   S->t_tmpl.as<int32_t>() + r0, S->t_nh.as<int32_t>() + r0, S->mt1_sum);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s));
	S->pool_used += (int64_t) used;
	// (their fragment rows are made beside the next batch)
	if((rc = mt1_frag_join(S))) return rc;
	S->frag_have = S->sink != nullptr; S->frag_r0 = r0; S->frag_n = nb; S->frag_max_len = B.max_len;
	return KMAHIP_OK;
}

static int session_map_one(kmahip_session *S, Batch &B) {
	hipStream_t s = 0;
	kmahip_db *db = S->db;
	kmahip_ws *ws = S->ws;
	const size_t D = db->info.DB_size;
	auto t = std::chrono::steady_clock::now();
	int rc;
	if(S->chain) {
		// stage 2 of the default mode: the chain finder on the batch's reads; its records -- in stream order -- go behind the records
		// of the batches before, with their bounds and the read whose header they carry
		if((rc = kmahip_chain_records_dev(db, ws, &B.tmp, &S->par, &S->cp, &B.cr))) { B.release(); return rc; }
		const int64_t m = B.cr.m;
		B.r0 = S->n; B.n = m;
		if(m) {
			const kmahip_reads &c = B.cr.d;
			if((rc = S->seq.ensure((size_t) (S->words + c.seq_words + 2) * 8, (size_t) S->words * 8, s)) || (rc = S->seq_off.ensure((size_t) (S->n + m + 1) * 8, (size_t) (S->n + 1) * 8, s)) ||
			   (rc = S->len.ensure((size_t) (S->n + m + 1) * 4, (size_t) S->n * 4, s)) || (rc = S->N.ensure((size_t) (S->nN + c.N_total + 1) * 4, (size_t) S->nN * 4, s)) ||
			   (rc = S->N_off.ensure((size_t) (S->n + m + 1) * 8, (size_t) (S->n + 1) * 8, s)) || (rc = S->qs.ensure((size_t) (S->n + m + 1) * 4, (size_t) S->n * 4, s)) ||
			   (rc = S->qe.ensure((size_t) (S->n + m + 1) * 4, (size_t) S->n * 4, s)) || (rc = S->rread.ensure((size_t) (S->n + m + 1) * 8, (size_t) S->n * 8, s))) { B.release(); return rc; }
			bool ok = true;
			ok = ok && (!c.seq_words || hipMemcpyAsync(S->seq.as<uint64_t>() + S->words, c.seq, (size_t) c.seq_words * 8, hipMemcpyDeviceToDevice, s) == hipSuccess);
			ok = ok && hipMemsetAsync(S->seq.as<uint64_t>() + S->words + c.seq_words, 0, 16, s) == hipSuccess;
			ok = ok && hipMemcpyAsync(S->seq_off.as<int64_t>() + S->n, c.seq_off, (size_t) (m + 1) * 8, hipMemcpyDeviceToDevice, s) == hipSuccess;
			ok = ok && hipMemcpyAsync(S->len.as<int32_t>() + S->n, c.len, (size_t) m * 4, hipMemcpyDeviceToDevice, s) == hipSuccess;
			ok = ok && (!c.N_total || hipMemcpyAsync(S->N.as<int32_t>() + S->nN, c.N, (size_t) c.N_total * 4, hipMemcpyDeviceToDevice, s) == hipSuccess);
			ok = ok && hipMemcpyAsync(S->N_off.as<int64_t>() + S->n, c.N_off, (size_t) (m + 1) * 8, hipMemcpyDeviceToDevice, s) == hipSuccess;
			ok = ok && hipMemcpyAsync(S->qs.as<int32_t>() + S->n, c.q_start, (size_t) m * 4, hipMemcpyDeviceToDevice, s) == hipSuccess;
			ok = ok && hipMemcpyAsync(S->qe.as<int32_t>() + S->n, c.q_end, (size_t) m * 4, hipMemcpyDeviceToDevice, s) == hipSuccess;
			ok = ok && hipMemcpyAsync(S->rread.as<int64_t>() + S->n, B.cr.o_read, (size_t) m * 8, hipMemcpyDeviceToDevice, s) == hipSuccess;
			if(!ok) { B.release(); kmahip_set_error("device copy failed"); return KMAHIP_EDEVICE; }
			const unsigned g1 = (unsigned) ((m + 1 + 255) / 256);
			if(S->words) hipLaunchKernelGGL(add_off_kernel, dim3(g1), dim3(256), 0, s, m + 1, S->seq_off.as<int64_t>() + S->n, S->words);
			if(S->nN) hipLaunchKernelGGL(add_off_kernel, dim3(g1), dim3(256), 0, s, m + 1, S->N_off.as<int64_t>() + S->n, S->nN);
			if(B.read0) hipLaunchKernelGGL(add_off_kernel, dim3(g1), dim3(256), 0, s, m, S->rread.as<int64_t>() + S->n, B.read0);
			if(hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { B.release(); kmahip_set_error("device copy failed"); return KMAHIP_EDEVICE; }
			B.c = B.cr.c; B.total = B.cr.n_T;
			S->n += m; S->words += c.seq_words; S->nN += c.N_total;
		}
		B.release_tmp();
	}
	const int64_t nb = B.n;
	kmahip_reads d{};
	d.n_reads = nb; d.seq = S->seq.as<uint64_t>(); d.seq_off = S->seq_off.as<int64_t>() + B.r0; d.len = S->len.as<int32_t>() + B.r0; d.N = S->N.as<int32_t>();
	d.N_off = S->N_off.as<int64_t>() + B.r0; d.seq_words = S->words; d.N_total = S->nN; d.max_len = B.max_len;
	if(S->chain && nb) { d.q_start = S->qs.as<int32_t>() + B.r0; d.q_end = S->qe.as<int32_t>() + B.r0; }
	if(S->chain && !nb) { S->batches.push_back(std::move(B)); S->ms_map += since(t); return KMAHIP_OK; }       // (a batch without a record)
	if(S->mt1) {
		rc = session_map_mt1(S, B, d);
		if(!rc) S->batches.push_back(std::move(B));
		S->ms_map += since(t);
		return rc;
	}
	// stage 2 (the candidate lists have no bound known in advance: two per read, redone with the exact size if short)
	if(!S->chain) {
	if((rc = dev_new(B.owned, (size_t) nb + 1, &B.c.rc_flag, false, s)) || (rc = dev_new(B.owned, (size_t) nb + 1, &B.c.flag, false, s)) || (rc = dev_new(B.owned, (size_t) nb + 1, &B.c.T_off, true, s))) { B.release(); return rc; }
	B.c.T_cap = 2 * nb + 4096;
	for(int attempt = 0;; ++attempt) {
		if((rc = dev_new(B.owned, (size_t) B.c.T_cap, &B.c.T, false, s)) || (rc = kmahip_launch_scan_se(db, ws, &d, &S->par, &B.c, s))) { B.release(); return rc; }
		if(hipStreamSynchronize(s) != hipSuccess) { B.release(); kmahip_set_error("stage 2 failed"); return KMAHIP_EDEVICE; }
		if(ws_status(ws, nullptr) == 1) {
			if(attempt >= 4) { B.release(); kmahip_set_error("internal candidate pool exhausted"); return KMAHIP_EOVERFLOW; }
			ws->pool_scale *= 2; ws->cap_reads = 0;
			continue;
		}
		if(hipMemcpy(&B.total, B.c.T_off + nb, 8, hipMemcpyDeviceToHost) != hipSuccess) { B.release(); kmahip_set_error("hipMemcpy failed"); return KMAHIP_EDEVICE; }
		if(B.total <= B.c.T_cap) break;
		if(attempt >= 6) { B.release(); kmahip_set_error("candidate lists keep growing"); return KMAHIP_EOVERFLOW; }
		B.c.T_cap = B.total + 1024;
	}
	}
	// stage 3a into vectors of the batch's own (a run that has to be repeated with more room for seeds starts them afresh), then added
	kmahip_hits &h = B.h;
	if((rc = dev_new(B.owned, (size_t) nb + 1, &h.n_hits, true, s)) || (rc = dev_new(B.owned, (size_t) nb + 1, &h.best_score, true, s)) || (rc = dev_new(B.owned, (size_t) nb + 1, &h.flag, true, s)) ||
	   (rc = dev_new(B.owned, (size_t) nb + 1, &h.rc, true, s)) || (rc = dev_new(B.owned, (size_t) B.total + 1, &h.tmpl, true, s)) || (rc = dev_new(B.owned, (size_t) B.total + 1, &h.score, true, s)) ||
	   (rc = dev_new(B.owned, (size_t) B.total + 1, &h.start, true, s)) || (rc = dev_new(B.owned, (size_t) B.total + 1, &h.end, true, s))) { B.release(); return rc; }
	h.alignment_scores = S->AS_batch; h.uniq_alignment_scores = S->AS_batch + D;
	for(;;) {
		if(hipMemsetAsync(S->AS_batch, 0, 2 * D * 8, s) != hipSuccess || (rc = kmahip_stage3a_se(db, ws, &d, &B.c, &S->par, &h, s))) { B.release(); return rc ? rc : KMAHIP_EDEVICE; }
		if(hipStreamSynchronize(s) != hipSuccess) { B.release(); kmahip_set_error("stage 3a failed"); return KMAHIP_EDEVICE; }
		if(ws_status(ws, nullptr) != 3) break;
		if(!grow_mem_cap(ws)) { B.release(); kmahip_set_error("seed (MEM) capacity per read/template pair exceeded"); return KMAHIP_EOVERFLOW; }
	}
	hipLaunchKernelGGL(add_u64_kernel, dim3((unsigned) ((2 * D + 255) / 256)), dim3(256), 0, s, (int64_t) (2 * D), (const unsigned long long *) S->AS_batch, (unsigned long long *) S->AS);
	if(hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { B.release(); kmahip_set_error("stage 3a failed"); return KMAHIP_EDEVICE; }
	h.alignment_scores = S->AS; h.uniq_alignment_scores = S->AS + D;
	const int64_t first_reads = S->batches.empty() ? B.n : 0;
	const int first_len = B.max_len;
	S->batches.push_back(std::move(B));
	// (what the finish needs once per process -- the traceback's scratch: gigabytes -- is made now, beside stage 1 of the next batch)
	if(first_reads && (rc = kmahip_trace_reserve(ws, first_len, std::max<int64_t>(first_reads, 262144)))) return rc;
	S->ms_map += since(t);
	return KMAHIP_OK;
}

// ---- the fragment rows of a run whose reads (or records, or fragments) and headers are in HBM: order, lengths and text on the device;
// the host compresses and writes. Items 0 .. n - 1 = the entries of W; d_name_idx (or NULL): the header an item carries; d_rank (or NULL:
// counted here): an item's position among the filed ones of the whole stream. order: 0 ConClave's (assemble_KMA's), 1 the stream's. pinned: three host buffers of text_chunk + 16 bytes made
// by the caller ahead of time, or NULL.
struct KmaFragSink {
	kmahip_gzstream *gz = nullptr;
	std::atomic<int> pending[3];
	int chunk_no = 0;
	// a stream of its own (the rows of one batch are made while the next batch is traced on the others), the template names once
	hipStream_t stream = nullptr;
	char *d_tn = nullptr;
	int64_t *d_tn_off = nullptr;
};
KmaFragSink *kmahip_frag_sink_open(const char *path) {
	KmaFragSink *k = new KmaFragSink();
	for(int x = 0; x < 3; ++x) k->pending[x].store(0);
	if(hipStreamCreateWithFlags(&k->stream, hipStreamNonBlocking) != hipSuccess) { k->stream = nullptr; delete k; kmahip_set_error("hipStreamCreate failed"); return nullptr; }
	k->gz = kmahip_gzstream_open(path);
	if(!k->gz) { (void) hipStreamDestroy(k->stream); delete k; return nullptr; }
	return k;
}
int kmahip_frag_sink_close(KmaFragSink *k) {
	if(!k) return KMAHIP_OK;
	const int rc = k->gz ? kmahip_gzstream_close(k->gz) : KMAHIP_OK;
	if(k->stream) (void) hipStreamDestroy(k->stream);
	if(k->d_tn) (void) hipFree(k->d_tn);
	if(k->d_tn_off) (void) hipFree(k->d_tn_off);
	delete k;
	return rc;
}

int kmahip_frag_write_dev(kmahip_db *db, const kmahip_reads *W, const char *d_names, const int64_t *d_name_off, const int64_t *d_name_idx, const int32_t *d_rc,
                          const int32_t *d_tmpl, const int32_t *d_nhits, const int32_t *d_stats, const int64_t *d_rank, int64_t mf, const char *path,
                          int64_t text_chunk, char **pinned, int64_t *n_rows_out, int order, KmaFragSink *sink) {
	if(sink && !pinned) { kmahip_set_error("a fragment file that stays open needs the caller's buffers"); return KMAHIP_EINVAL; }
	const int64_t n = W->n_reads;
	const size_t D = db->info.DB_size;
	hipStream_t s = sink ? sink->stream : 0;
	int rc;
	auto t = std::chrono::steady_clock::now();
	if((rc = kmahip_db_load_names(db))) return rc;
	DevBlock B;
	B.expect((size_t) n * 72 + (64u << 20));
	char *own[3] = {nullptr, nullptr, nullptr};
	struct Own { char **o; ~Own() { for(int x = 0; x < 3; ++x) if(o[x]) (void) hipHostFree(o[x]); } } own_guard{own};
	char *h_text_arr[3] = {pinned ? pinned[0] : nullptr, pinned ? pinned[1] : nullptr, pinned ? pinned[2] : nullptr};
	char **h_text_in = h_text_arr;
	if(text_chunk <= 0) text_chunk = 64ll << 20;
	if(!pinned) {
		for(int x = 0; x < 3; ++x) { if(hipHostMalloc((void **) &own[x], (size_t) text_chunk + 16, hipHostMallocDefault) != hipSuccess) { own[x] = nullptr; kmahip_set_error("hipHostMalloc failed"); return KMAHIP_ENOMEM; } h_text_arr[x] = own[x]; }
	}
	const std::string prefix_path(path);
	int64_t *filed = nullptr, *kept = nullptr, *frank = nullptr, *kscan = nullptr, *vals = nullptr, *vals2 = nullptr, *row_off = nullptr, *row_len = nullptr;
	unsigned long long *keys = nullptr, *keys2 = nullptr;
	if((rc = B.get((size_t) n + 1, &filed)) || (rc = B.get((size_t) n + 1, &kept)) || (rc = B.get((size_t) n + 1, &frank)) || (rc = B.get((size_t) n + 1, &kscan)) ||
	   (rc = B.get((size_t) n + 1, &keys)) || (rc = B.get((size_t) n + 1, &keys2)) || (rc = B.get((size_t) n + 1, &vals)) || (rc = B.get((size_t) n + 1, &vals2))) return rc;
	hipLaunchKernelGGL(row_flags_kernel, dim3((unsigned) ((n + 256) / 256)), dim3(256), 0, s, n, d_tmpl, d_stats, filed, kept);
	HIP_TRY(hipGetLastError());
	if((rc = scan_i64(B, filed, frank, (size_t) n + 1, s)) || (rc = scan_i64(B, kept, kscan, (size_t) n + 1, s))) return rc;
	const int64_t *use_rank = d_rank ? d_rank : frank;
	int64_t n_frag_rows = 0;
	HIP_TRY(hipMemcpyAsync(&n_frag_rows, kscan + n, 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	if(n_rows_out) *n_rows_out = n_frag_rows;
	kmahip_gzstream *gz = sink ? sink->gz : kmahip_gzstream_open(prefix_path.c_str());
	if(!gz) return KMAHIP_EIO;
	struct Closer { kmahip_gzstream *&g; bool own; ~Closer() { if(g && own) (void) kmahip_gzstream_close(g); } } closer{gz, sink == nullptr};
	if(n_frag_rows > 0) {
		hipLaunchKernelGGL(row_keys_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, n, d_tmpl, kscan, use_rank, mf, order, keys, vals);
		HIP_TRY(hipGetLastError());
		{
			size_t tmp_bytes = 0;
			if(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, keys2, vals, vals2, (size_t) n, 0, 64, s) != hipSuccess) { kmahip_set_error("rocprim::radix_sort_pairs (size query) failed"); return KMAHIP_EDEVICE; }
			char *tmp = nullptr;
			if((rc = B.get(tmp_bytes, &tmp))) return rc;
			if(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, vals, vals2, (size_t) n, 0, 64, s) != hipSuccess) { kmahip_set_error("rocprim::radix_sort_pairs failed"); return KMAHIP_EDEVICE; }
		}
		// template names on the device
		const char *d_tn = sink ? sink->d_tn : nullptr;
		const int64_t *d_tn_off = sink ? sink->d_tn_off : nullptr;
		if(!d_tn) {
			std::vector<int64_t> tn_off(D + 1, 0);
			std::string tn;
			for(size_t tt = 1; tt < D; ++tt) { if(tt - 1 < db->h_names.size()) tn += db->h_names[tt - 1]; tn_off[tt] = (int64_t) tn.size(); }
			tn_off[D] = (int64_t) tn.size();
			if(sink) {          // (kept with the file)
				if(hipMalloc((void **) &sink->d_tn, tn.size() + 1) != hipSuccess || hipMalloc((void **) &sink->d_tn_off, (D + 1) * 8) != hipSuccess) { kmahip_set_error("hipMalloc failed"); return KMAHIP_ENOMEM; }
				HIP_TRY(hipMemcpy(sink->d_tn, tn.data(), tn.size(), hipMemcpyHostToDevice));
				HIP_TRY(hipMemcpy(sink->d_tn_off, tn_off.data(), (D + 1) * 8, hipMemcpyHostToDevice));
				d_tn = sink->d_tn; d_tn_off = sink->d_tn_off;
			}
			else if((rc = B.up(tn.data(), tn.size(), 1, &d_tn)) || (rc = B.up(tn_off.data(), D + 1, 0, &d_tn_off))) return rc;
		}
		if((rc = B.get((size_t) n_frag_rows + 1, &row_len)) || (rc = B.get((size_t) n_frag_rows + 1, &row_off))) return rc;
		RowArgs A{};
		A.seq = W->seq; A.seq_off = W->seq_off; A.N_off = W->N_off; A.name_off = d_name_off; A.len = W->len; A.N = W->N; A.rc = d_rc; A.tmpl = d_tmpl; A.n_hits = d_nhits;
		A.stats = d_stats; A.names = d_names; A.tnames = d_tn; A.tname_off = d_tn_off; A.row_read = vals2; A.row_off = row_len;
		A.name_idx = d_name_idx;
		hipLaunchKernelGGL(row_len_kernel, dim3((unsigned) ((n_frag_rows + 256) / 256)), dim3(256), 0, s, A, n_frag_rows);
		HIP_TRY(hipGetLastError());
		if((rc = scan_i64(B, row_len, row_off, (size_t) n_frag_rows + 1, s))) return rc;
		A.row_off = row_off;
		int64_t text_bytes = 0;
		HIP_TRY(hipMemcpyAsync(&text_bytes, row_off + n_frag_rows, 8, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		// blocks of rows of about 4 MB of text (a gzip member each), chunks of blocks of at most CHUNK bytes through two text buffers
		const int64_t avg = std::max<int64_t>(1, text_bytes / n_frag_rows);
		const int64_t rows_per_block = std::max<int64_t>(16, std::min<int64_t>(1 << 16, (4 << 20) / avg));
		const int64_t n_blocks = (n_frag_rows + rows_per_block - 1) / rows_per_block;
		int64_t *d_boff = nullptr;
		if((rc = B.get((size_t) n_blocks + 1, &d_boff))) return rc;
		hipLaunchKernelGGL(row_blocks_kernel, dim3((unsigned) ((n_blocks + 256) / 256)), dim3(256), 0, s, n_blocks, rows_per_block, n_frag_rows, row_off, d_boff);
		HIP_TRY(hipGetLastError());
		std::vector<int64_t> boff((size_t) n_blocks + 1);
		HIP_TRY(hipMemcpyAsync(boff.data(), d_boff, ((size_t) n_blocks + 1) * 8, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		int64_t max_block = 0;
		for(int64_t b = 0; b < n_blocks; ++b) max_block = std::max(max_block, boff[(size_t) b + 1] - boff[(size_t) b]);
		const int64_t CHUNK = std::max<int64_t>(text_chunk, max_block);
		constexpr int NBUF = 3;
		char *d_text[2] = {nullptr, nullptr};
		char **h_text = h_text_in;
		std::atomic<int> pending_own[NBUF];
		for(int x = 0; x < NBUF; ++x) pending_own[x].store(0);
		std::atomic<int> *pending = sink ? sink->pending : pending_own;
		for(int x = 0; x < 2; ++x) if((rc = B.get((size_t) CHUNK + 16, &d_text[x]))) return rc;
		if(CHUNK > text_chunk) {          // (a single block of rows longer than the buffers made at the start: rows of very long reads)
			// (the caller's buffers stay as they are: three larger ones of our own for this file)
			for(int x = 0; x < NBUF; ++x) { if(own[x]) (void) hipHostFree(own[x]); own[x] = nullptr; HIP_TRY(hipHostMalloc((void **) &own[x], (size_t) CHUNK + 16, hipHostMallocDefault)); h_text[x] = own[x]; }
		}
		int chunk_no = sink ? sink->chunk_no : 0;
		const int chunk_first = chunk_no;
		const bool dbg = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
		double ms_prep = since(t), ms_fmt = 0, ms_wait = 0, ms_copy = 0;
		auto lap = std::chrono::steady_clock::now();
		for(int64_t b0 = 0; b0 < n_blocks; ++chunk_no) {
			int64_t b1 = b0 + 1;
			while(b1 < n_blocks && boff[(size_t) b1 + 1] - boff[(size_t) b0] <= CHUNK) ++b1;
			const int64_t r0 = b0 * rows_per_block, r1 = std::min(n_frag_rows, b1 * rows_per_block), bytes = boff[(size_t) b1] - boff[(size_t) b0];
			char *dt = d_text[(chunk_no - chunk_first) & 1];
			const int hb = chunk_no % NBUF;
			hipLaunchKernelGGL(row_format_kernel, dim3((unsigned) ((r1 - r0 + 255) / 256)), dim3(256), 0, s, A, r0, r1, boff[(size_t) b0], dt);
			HIP_TRY(hipGetLastError());
			if(dbg) { HIP_TRY(hipStreamSynchronize(s)); ms_fmt += since(lap); }
			while(pending[hb].load() > 0) std::this_thread::yield();          // (the buffer's blocks of three chunks ago are still being compressed)
			if(dbg) ms_wait += since(lap);
			HIP_TRY(hipMemcpyAsync(h_text[hb], dt, (size_t) bytes, hipMemcpyDeviceToHost, s));
			HIP_TRY(hipStreamSynchronize(s));
			if(dbg) ms_copy += since(lap);
			pending[hb].store((int) (b1 - b0));
			for(int64_t b = b0; b < b1; ++b) kmahip_gzstream_submit(gz, h_text[hb] + (boff[(size_t) b] - boff[(size_t) b0]), (size_t) (boff[(size_t) b + 1] - boff[(size_t) b]), &pending[hb]);
			b0 = b1;
		}
		if(sink) {
			sink->chunk_no = chunk_no;
			// (the file stays open and the threads go on compressing out of the caller's buffers; buffers of our own go with this call)
			if(CHUNK > text_chunk) for(int x = 0; x < NBUF; ++x) while(pending[x].load() > 0) std::this_thread::yield();
		} else {
			kmahip_gzstream *g = gz;
			gz = nullptr;
			if((rc = kmahip_gzstream_close(g))) return rc;          // (before the pinned buffers go)
		}
		if(dbg) fprintf(stderr, "[kmahip] session: fragment rows: %lld rows, %lld bytes of text in %d chunks of %lld blocks; order + lengths + buffers %.1f ms, formatting %.1f, waiting for a free buffer %.1f, copies %.1f, draining the writer %.1f\n",
		                (long long) n_frag_rows, (long long) text_bytes, chunk_no, (long long) n_blocks, ms_prep, ms_fmt, ms_wait, ms_copy, since(lap));
	} else if(!sink) {
		kmahip_gzstream *g = gz;
		gz = nullptr;
		if((rc = kmahip_gzstream_close(g))) return rc;
	}
	(void) t;
	return KMAHIP_OK;
}

// ConClave, statistics, traceback, pile-up, consensus and the three files. ms[8]: uploads (summed over the batches), stages 2 + 3a
// (summed), ConClave + statistics, traceback, pile-up + consensus, .res + .fsa, fragment rows, (unused).
extern "C" int kmahip_session_finish(kmahip_session *S, const char *out_prefix, int write_fsa, int write_frag, int64_t *n_reads, int64_t *n_rows_out, double ms[8]) {
	if(!S || !out_prefix || !ms) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	kmahip_db *db = S->db;
	kmahip_ws *ws = S->ws;
	const kmahip_params *p = &S->par;
	const size_t D = db->info.DB_size;
	const int64_t n = S->n;
	const int64_t mf = S->opts.max_frag > 0 ? S->opts.max_frag : 1000000;
	hipStream_t s = 0;
	for(int i = 0; i < 8; ++i) ms[i] = 0;
	ms[0] = S->ms_upload; ms[1] = S->ms_map;
	if(n_reads) *n_reads = S->n_reads;
	if(n_rows_out) *n_rows_out = 0;
	auto t = std::chrono::steady_clock::now();
	int rc;
	if((rc = kmahip_session_map(S))) return rc;
	ms[1] = S->ms_map;
	if((rc = kmahip_db_load_names(db))) return rc;
	DevBlock B;
	B.expect((size_t) n * 120 + (64u << 20));
	// (the arrays are read one element past the end by some kernels: make sure those exist when nothing was ever added)
	if((rc = S->seq.ensure(16, 0, s)) || (rc = S->seq_off.ensure(16, 0, s)) || (rc = S->len.ensure(8, 0, s)) || (rc = S->N.ensure(8, 0, s)) || (rc = S->N_off.ensure(16, 0, s)) ||
	   (rc = S->names.ensure(8, 0, s)) || (rc = S->name_off.ensure(16, 0, s))) return rc;
	if(n == 0) { HIP_TRY(hipMemsetAsync(S->seq_off.p, 0, 16, s)); HIP_TRY(hipMemsetAsync(S->N_off.p, 0, 16, s)); HIP_TRY(hipMemsetAsync(S->name_off.p, 0, 16, s)); }
	kmahip_reads W{};
	W.n_reads = n; W.seq = S->seq.as<uint64_t>(); W.seq_off = S->seq_off.as<int64_t>(); W.len = S->len.as<int32_t>(); W.N = S->N.as<int32_t>(); W.N_off = S->N_off.as<int64_t>();
	W.seq_words = S->words; W.N_total = S->nN; W.max_len = S->max_len;
	if(S->chain && n) { W.q_start = S->qs.as<int32_t>(); W.q_end = S->qe.as<int32_t>(); }

	if(S->pe) {
		// paired input: the whole run on what the batches left in HBM (kmahip_run_pe's stages; pipeline.hip), then the text files
		const std::string prefix(out_prefix);
		const std::string frag = prefix + ".frag.gz";
		kmahip_read_batch hb;
		session_pe_view(S, 0, n, &hb);
		int64_t n_frag_rows = 0;
		KmaPeDev pd{S->names.as<char>(), S->name_off.as<int64_t>(), S->h_text, S->text_chunk, &n_frag_rows};
		std::vector<kmahip_res_row> rows(D);
		std::vector<int64_t> a_cover(D, 0), a_len(D, 0), a_depth(D, 0), a_asm(D, 0), c_off(D, -1);
		int64_t tbases = 0;
		for(size_t tt = 1; tt < D; ++tt) tbases += db->h_tlen[tt];
		std::vector<char> cons((size_t) (4 * tbases + 4 * (int64_t) D + (1 << 20)));
		kmahip_run run;
		memset(&run, 0, sizeof run);
		run.rows = rows.data(); run.rows_cap = (int64_t) D;
		run.assembly.cover = a_cover.data(); run.assembly.aln_len = a_len.data(); run.assembly.depth = a_depth.data(); run.assembly.asm_len = a_asm.data();
		run.assembly.consensus = cons.data(); run.assembly.consensus_off = c_off.data(); run.assembly.consensus_cap = (int64_t) cons.size(); run.assembly.consensus_used = 0;
		run.caller = S->opts.caller | (S->opts.ref_fsa == 2 ? 8 : 0) | (S->opts.write_aln ? 32 : 0); run.sig90 = S->opts.sig90; run.support = S->opts.support;
		if((rc = kmahip_run_pe_resident(db, ws, &hb, &pd, p, S->opts.evalue, S->opts.bcd, S->opts.max_frag, write_frag ? frag.c_str() : nullptr, &run))) return rc;
		(void) since(t);
		ms[1] += run.ms[0] + run.ms[1]; ms[2] = run.ms[2]; ms[3] = run.ms[3]; ms[4] = run.ms[4]; ms[6] = run.ms[5];
		if((rc = kmahip_write_res_fsa(db, (prefix + ".res").c_str(), write_fsa ? (prefix + ".fsa").c_str() : nullptr, true, rows.data(), run.n_rows, nullptr, 0, a_cover.data(), a_len.data(),
		                              a_depth.data(), cons.data(), c_off.data(), S->opts.ID_t > 0 ? S->opts.ID_t : 1.0, S->opts.Depth_t, S->opts.ref_fsa, S->opts.write_aln ? (prefix + ".aln").c_str() : nullptr))) return rc;
		ms[5] = since(t);
		if(n_rows_out) *n_rows_out = n_frag_rows;
		return KMAHIP_OK;
	}

	if(S->mt1) {
		// `-Mt1`: the tracebacks are there (kmahip_session_map); the `.res` row, the pile-up in stream order, the files
		const int32_t tmpl = S->mt1;
		const int t_len = db->h_tlen[(size_t) tmpl];
		unsigned long long score = 0;
		if((rc = S->t_stats.ensure(48, 0, s)) || (rc = S->t_off.ensure(16, 0, s)) || (rc = S->t_nops.ensure(8, 0, s)) || (rc = S->t_rc.ensure(8, 0, s)) || (rc = S->t_pool.ensure(64, 0, s)) ||
		   (rc = S->t_tmpl.ensure(8, 0, s)) || (rc = S->t_nh.ensure(8, 0, s))) return rc;
		const int32_t *d_tmpl = S->t_tmpl.as<int32_t>(), *d_nh = S->t_nh.as<int32_t>();
		mt1_frag_start(S);          // (the last batch's rows, beside the pile-up)
		struct Join { kmahip_session *S; ~Join() { (void) mt1_frag_join(S); } } join_guard{S};
		HIP_TRY(hipMemcpy(&score, S->mt1_sum, 8, hipMemcpyDeviceToHost));
		kmahip_res_row row;
		memset(&row, 0, sizeof row);
		row.template_id = tmpl; row.template_length = t_len; row.score = score; row.expected = 0;
		row.q_value = (double) score; row.p_value = kmahip_p_chisqr((long double) score);
		row.significant = kmahip_cmp(row.p_value <= S->opts.evalue && score > 0, (double) score >= p->scoreT * t_len);     // mt1.c:419
		ms[2] = since(t);
		kmahip_traces tr{};
		tr.stats = S->t_stats.as<int32_t>(); tr.ops_off = S->t_off.as<int64_t>(); tr.n_ops = S->t_nops.as<int32_t>(); tr.ops = S->t_pool.as<uint32_t>(); tr.ops_cap = (int64_t) (S->t_pool.cap / 4);
		std::vector<int64_t> a_cover(D, 0), a_len(D, 0), a_depth(D, 0), a_asm(D, 0), c_off(D, -1);
		std::vector<char> cons((size_t) (4 * (int64_t) t_len + 4 * (int64_t) D + (1 << 20)));
		kmahip_assembly asmb{};
		asmb.cover = a_cover.data(); asmb.aln_len = a_len.data(); asmb.depth = a_depth.data(); asmb.asm_len = a_asm.data();
		asmb.consensus = cons.data(); asmb.consensus_off = c_off.data(); asmb.consensus_cap = (int64_t) cons.size(); asmb.consensus_used = 0;
		if(n && score) {
			kmahip_assemble_opts ao = {mf, S->opts.evalue, S->opts.bcd, 1, S->opts.caller | (S->opts.ref_fsa == 2 ? 8 : 0) | (S->opts.write_aln ? 32 : 0), S->opts.sig90, nullptr, S->opts.support};
			if((rc = kmahip_assemble2_dev(db, ws, &W, S->t_rc.as<int32_t>(), d_tmpl, &tr, &ao, &asmb))) return rc;
		}
		ms[4] = since(t);
		const std::string prefix(out_prefix);
		if((rc = kmahip_write_res_fsa(db, (prefix + ".res").c_str(), write_fsa ? (prefix + ".fsa").c_str() : nullptr, true, &row, 1, nullptr, 0, a_cover.data(), a_len.data(),
		                              a_depth.data(), cons.data(), c_off.data(), S->opts.ID_t > 0 ? S->opts.ID_t : 1.0, S->opts.Depth_t, S->opts.ref_fsa, S->opts.write_aln ? (prefix + ".aln").c_str() : nullptr))) return rc;
		ms[5] = since(t);
		if(S->sink) {          // (written batch by batch: what is left is the end of the file)
			if((rc = mt1_frag_join(S))) return rc;
			KmaFragSink *k = S->sink;
			S->sink = nullptr;
			if((rc = kmahip_frag_sink_close(k))) return rc;
			if(n_rows_out) *n_rows_out = S->sink_rows;
			ms[6] = since(t);
			ms[7] = S->ms_frag;
			return KMAHIP_OK;
		}
		if(!write_frag) return KMAHIP_OK;
		int64_t n_frag_rows = 0;
		if((rc = kmahip_frag_write_dev(db, &W, S->names.as<char>(), S->name_off.as<int64_t>(), nullptr, S->t_rc.as<int32_t>(), d_tmpl, d_nh, tr.stats, nullptr, mf,
		                               (prefix + ".frag.gz").c_str(), S->text_chunk, S->h_text, &n_frag_rows, 1))) return rc;
		if(n_rows_out) *n_rows_out = n_frag_rows;
		ms[6] = since(t);
		return KMAHIP_OK;
	}

	// stage 3b per batch on the finished vectors, the `.res` statistics
	kmahip_conclave cc{};
	int32_t *rc_all = nullptr, *nh_all = nullptr;
	if((rc = B.get((size_t) n + 1, &cc.tmpl, true)) || (rc = B.get((size_t) n + 1, &cc.start, true)) || (rc = B.get((size_t) n + 1, &cc.end, true)) || (rc = B.get(D, &cc.w_scores, true)) ||
	   (rc = B.get((size_t) n + 1, &rc_all, true)) || (rc = B.get((size_t) n + 1, &nh_all, true))) return rc;
	for(Batch &b : S->batches) {
		if(!b.n) continue;
		kmahip_reads d = W;
		d.n_reads = b.n; d.seq_off = W.seq_off + b.r0; d.len = W.len + b.r0; d.N_off = W.N_off + b.r0; d.max_len = b.max_len;
		if(W.q_start) { d.q_start = W.q_start + b.r0; d.q_end = W.q_end + b.r0; }
		kmahip_conclave cb = cc;
		cb.tmpl = cc.tmpl + b.r0; cb.start = cc.start + b.r0; cb.end = cc.end + b.r0;
		if((rc = kmahip_conclave_se_dev(db, ws, &d, &b.c, &b.h, &cb, s))) return rc;
		HIP_TRY(hipMemcpyAsync(rc_all + b.r0, b.h.rc, (size_t) b.n * 4, hipMemcpyDeviceToDevice, s));
		HIP_TRY(hipMemcpyAsync(nh_all + b.r0, b.h.n_hits, (size_t) b.n * 4, hipMemcpyDeviceToDevice, s));
	}
	std::vector<uint64_t> w(D);
	HIP_TRY(hipMemcpy(w.data(), cc.w_scores, D * 8, hipMemcpyDeviceToHost));
	std::vector<kmahip_res_row> rows(D);
	int64_t n_rows = 0;
	if((rc = kmahip_res_rows(db, w.data(), S->opts.evalue, p->scoreT, rows.data(), (int64_t) D, &n_rows))) return rc;
	std::vector<uint8_t> ok(D + 8, 0);
	for(int64_t r = 0; r < n_rows; ++r) ok[(size_t) rows[(size_t) r].template_id] = (uint8_t) rows[(size_t) r].significant;
	const uint8_t *d_ok = nullptr;
	if((rc = B.up(ok.data(), D + 8, 0, &d_ok))) return rc;
	ms[2] = since(t);

	// the traceback over everything in one launch (its second pass lasts as long as its slowest lane: once, not once per batch)
	kmahip_traces tr{};
	if((rc = B.get((size_t) 10 * n + 10, &tr.stats, true)) || (rc = B.get((size_t) n + 1, &tr.ops_off, true)) || (rc = B.get((size_t) n + 1, &tr.n_ops, true))) return rc;
	for(Batch &b : S->batches) b.release();
	DevArr pool;
	// (runs: a handful per short read; long reads with their errors leave one every few bases)
	if((rc = pool.ensure((size_t) (6 * n + (1 << 20) + (S->max_len > 1024 ? S->words * 32 / 3 : 0)) * 4, 0, s))) return rc;
	for(int attempt = 0; n; ++attempt) {
		tr.ops = pool.as<uint32_t>(); tr.ops_cap = (int64_t) (pool.cap / 4);
		if((rc = kmahip_launch_trace(db, ws, &W, rc_all, cc.tmpl, d_ok, p, &tr, s))) return rc;
		HIP_TRY(hipStreamSynchronize(s));
		unsigned long long used = 0;
		const int st = ws_status(ws, &used);
		if(st == 2 || (int64_t) used > tr.ops_cap) {
			if(attempt >= 3) { kmahip_set_error("alignment run pool: %llu runs needed", used); return KMAHIP_EOVERFLOW; }
			if((rc = pool.ensure((size_t) ((int64_t) used + (1 << 20)) * 4, 0, s))) return rc;
			continue;
		}
		if(st == 16 && grow_mem_cap(ws)) { --attempt; continue; }
		if(st) { kmahip_set_error("trace stage: a read needs more scratch than the workspace holds (status %d)", st); return KMAHIP_EDEVICE; }
		break;
	}
	tr.ops = pool.as<uint32_t>(); tr.ops_cap = (int64_t) (pool.cap / 4);
	ms[3] = since(t);

	// stage 3c per template over everything
	std::vector<int64_t> a_cover(D, 0), a_len(D, 0), a_depth(D, 0), a_asm(D, 0), c_off(D, -1);
	int64_t tbases = 0;
	for(size_t tt = 1; tt < D; ++tt) tbases += db->h_tlen[tt];
	std::vector<char> cons((size_t) (4 * tbases + 4 * (int64_t) D + (1 << 20)));
	kmahip_assembly asmb{};
	asmb.cover = a_cover.data(); asmb.aln_len = a_len.data(); asmb.depth = a_depth.data(); asmb.asm_len = a_asm.data();
	asmb.consensus = cons.data(); asmb.consensus_off = c_off.data(); asmb.consensus_cap = (int64_t) cons.size(); asmb.consensus_used = 0;
	if(n) {
		kmahip_assemble_opts ao = {mf, S->opts.evalue, S->opts.bcd, 0, S->opts.caller | (S->opts.ref_fsa == 2 ? 8 : 0) | (S->opts.write_aln ? 32 : 0), S->opts.sig90, nullptr, S->opts.support};
		if((rc = kmahip_assemble2_dev(db, ws, &W, rc_all, cc.tmpl, &tr, &ao, &asmb))) return rc;
	}
	ms[4] = since(t);
	const std::string prefix(out_prefix);
	// `.res`, `.fsa` and `.aln` (one host thread formats them: 45 ms for 5 k genes) beside the fragment rows
	int rc_text = KMAHIP_OK;
	std::string err_text;
	double ms_text = 0;
	std::thread text([&]() {
		auto tt = std::chrono::steady_clock::now();
		rc_text = kmahip_write_res_fsa(db, (prefix + ".res").c_str(), write_fsa ? (prefix + ".fsa").c_str() : nullptr, true, rows.data(), n_rows, nullptr, 0, a_cover.data(), a_len.data(),
		                               a_depth.data(), cons.data(), c_off.data(), S->opts.ID_t > 0 ? S->opts.ID_t : 1.0, S->opts.Depth_t, S->opts.ref_fsa, S->opts.write_aln ? (prefix + ".aln").c_str() : nullptr);
		if(rc_text) err_text = kmahip_last_error();          // (the message is the thread's)
		ms_text = since(tt);
	});
	struct Join { std::thread &th; ~Join() { if(th.joinable()) th.join(); } } join_text{text};
	if(write_frag) {
		// ---- the fragment rows: order, lengths and text on the device; the host compresses and writes (kmahip_frag_write_dev)
		int64_t n_frag_rows = 0;
		if((rc = kmahip_frag_write_dev(db, &W, S->names.as<char>(), S->name_off.as<int64_t>(), S->chain ? S->rread.as<int64_t>() : nullptr, rc_all, cc.tmpl, nh_all, tr.stats, nullptr, mf,
		                               (prefix + ".frag.gz").c_str(), S->text_chunk, S->h_text, &n_frag_rows))) return rc;
		if(n_rows_out) *n_rows_out = n_frag_rows;
	}
	text.join();
	if(rc_text) { kmahip_set_error("%s", err_text.c_str()); return rc_text; }
	ms[5] = ms_text;
	ms[6] = since(t);
	return KMAHIP_OK;
}
