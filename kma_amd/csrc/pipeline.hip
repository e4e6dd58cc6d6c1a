// pipeline.hip -- kmahip_run_se: the single-end `-1t1` run of one batch in one call (host buffers in, per-template results
// out), built from the same launchers as the stage-wise entry points; what runKMA does between its input stream and the
// `.res` / consensus output (runkma.c:104-900), minus the files.
#include "pipeline_util.h"
#include <mutex>

// ---- the large device blocks kept between runs (pipeline_util.h) -------------------------------------------------------------------
namespace {
struct DevCache {
	std::mutex m;
	std::vector<std::pair<void *, size_t>> kept;
	size_t total = 0;
	static constexpr size_t CAP = 64ull << 30;      // bytes kept at most
	static constexpr size_t MAXN = 12;               // blocks kept at most
} g_devcache;
}
void *kmahip_devcache_take(size_t bytes, size_t *got) {
	std::lock_guard<std::mutex> lk(g_devcache.m);
	int best = -1;
	for(size_t i = 0; i < g_devcache.kept.size(); ++i) {
		const size_t b = g_devcache.kept[i].second;
		if(b >= bytes && b <= 2 * bytes && (best < 0 || b < g_devcache.kept[(size_t) best].second)) best = (int) i;
	}
	if(best < 0) return nullptr;
	void *p = g_devcache.kept[(size_t) best].first;
	*got = g_devcache.kept[(size_t) best].second;
	g_devcache.total -= *got;
	g_devcache.kept.erase(g_devcache.kept.begin() + best);
	return p;
}
void kmahip_devcache_give(void *p, size_t bytes) {
	if(!p) return;
	{
		std::lock_guard<std::mutex> lk(g_devcache.m);
		if(!getenv("KMAHIP_NO_DEVCACHE") && bytes <= DevCache::CAP) {
			// (room is made by releasing the smallest blocks kept: the large ones are the expensive ones to get back)
			while(!g_devcache.kept.empty() && (g_devcache.total + bytes > DevCache::CAP || g_devcache.kept.size() >= DevCache::MAXN)) {
				size_t s = 0;
				for(size_t i = 1; i < g_devcache.kept.size(); ++i) if(g_devcache.kept[i].second < g_devcache.kept[s].second) s = i;
				(void) hipFree(g_devcache.kept[s].first);
				g_devcache.total -= g_devcache.kept[s].second;
				g_devcache.kept.erase(g_devcache.kept.begin() + (long) s);
			}
			g_devcache.kept.push_back({p, bytes});
			g_devcache.total += bytes;
			return;
		}
	}
	(void) hipFree(p);
}
void kmahip_devcache_flush() {
	std::lock_guard<std::mutex> lk(g_devcache.m);
	for(auto &b : g_devcache.kept) (void) hipFree(b.first);
	g_devcache.kept.clear();
	g_devcache.total = 0;
}
#include <functional>
#include <memory>


// ---- `-mem_mode` (runKMA_MEM, runkma.c:910-1250): no alignment before ConClave. A stage-2 record IS the frag_raw record -- its template
// list as the hits, each from 0 to the template's length, the record's k-mer score as the read score, added to the ConClave vectors
// (update_Scores_MEM, updatescores.c:31-67); ConClave, the `.res` statistics, and the traceback of stage 3c against the chosen template
// follow as ever. One setting per process, like the reference's choice of runKMA_MEM over runKMA (kma.c:1619-1623).
static int g_mem_mode = 0;
extern "C" int kmahip_set_mem_mode(int on) { g_mem_mode = on != 0; return KMAHIP_OK; }
int kmahip_mem_mode() { return g_mem_mode; }

namespace {
__global__ __launch_bounds__(256) void mem_hits_kernel(int64_t n, int k, const int32_t *len, const int32_t *rc_flag, const int32_t *flag, const int64_t *T_off, const int32_t *T,
                                                        const int32_t *tlen, kmahip_hits h, unsigned long long *AS, unsigned long long *US) {
	const int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(r >= n) return;
	const int64_t o = T_off[r], e = T_off[r + 1];
	const int fl = flag[r];
	h.flag[r] = fl;
	if(h.rc) h.rc[r] = 0;
	if(e == o || len[r] < k) { h.n_hits[r] = 0; h.best_score[r] = 0; return; }          // (runkma.c:1101: a read shorter than k leaves no record)
	const int score = abs(rc_flag[r]);
	for(int64_t t = o; t < e; ++t) {
		const int tm = T[t];
		h.tmpl[t] = tm; h.score[t] = score; h.start[t] = 0; h.end[t] = tlen[abs(tm)];
		if(AS) atomicAdd(&AS[abs(tm)], (unsigned long long) score);
	}
	if(e - o == 1 && US) atomicAdd(&US[abs(T[o])], (unsigned long long) score);
	h.n_hits[r] = (int32_t) (e - o); h.best_score[r] = score;
	if(h.rc) h.rc[r] = (fl & 16) != 0;          // (the record holds the read as stage 2 passed it on)
}
// the records of a paired stream (runkma.c:1090-1134, update_Scores_pe_MEM updatescores.c:69-107): a couple -- a first record without a
// list, then its mate with the templates -- is one frag_raw record with both mates' scores added up (negated there: two reads); any other
// record is a single one. Outputs in the layout of kmahip_launch_align_pe (pe_records_kernel reads them): kind 1 = couple, hits at the
// second slot's list; 0 = every present record by itself.
__global__ __launch_bounds__(256) void mem_hits_pe_kernel(int64_t np, int k, const int32_t *len, kmahip_pe_recs R, const int32_t *tlen, kmahip_hits h, int32_t *kind,
                                                           unsigned long long *AS, unsigned long long *US) {
	const int64_t j = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(j >= np) return;
	const int64_t r0 = 2 * j, r1 = r0 + 1;
	int64_t nl[2];
	int L[2];
	for(int x = 0; x < 2; ++x) {
		const int64_t r = r0 + x;
		const bool here = R.mate[r] >= 0;
		h.n_hits[r] = 0; h.best_score[r] = 0; h.flag[r] = R.flag[r];
		if(h.rc) h.rc[r] = here && R.rc[r] != 0;
		nl[x] = here ? R.R_off[r + 1] - R.R_off[r] : 0;
		L[x] = here ? len[2 * j + R.mate[r]] : 0;
	}
	auto file = [&](int64_t o, int64_t n, int score) {
		for(int64_t t = o; t < o + n; ++t) {
			const int tm = R.T[t];
			h.tmpl[t] = tm; h.score[t] = score; h.start[t] = 0; h.end[t] = tlen[abs(tm)];
			if(AS) atomicAdd(&AS[abs(tm)], (unsigned long long) score);
		}
		if(n == 1 && US) atomicAdd(&US[abs(R.T[o])], (unsigned long long) score);
	};
	int kd = 0;
	if(R.mate[r0] >= 0 && R.mate[r1] >= 0 && nl[0] == 0 && nl[1] > 0) {
		if(L[0] >= k) {
			const int s0 = abs(R.rc_flag[r0]), s1 = abs(R.rc_flag[r1]);
			if(s1 && L[1] >= k) {
				kd = 1;
				file(R.R_off[r1], nl[1], s0 + s1);
				h.n_hits[r0] = h.n_hits[r1] = (int32_t) nl[1]; h.best_score[r0] = h.best_score[r1] = s0 + s1;
			} else {          // (the mate too short for a k-mer: the first record is filed singly, with the list its mate brought)
				kd = 3;
				file(R.R_off[r1], nl[1], s0);
				h.n_hits[r0] = (int32_t) nl[1]; h.best_score[r0] = s0;
			}
		}
	} else {
		for(int x = 0; x < 2; ++x) {
			const int64_t r = r0 + x;
			if(nl[x] == 0 || L[x] < k) continue;
			const int s = abs(R.rc_flag[r]);
			file(R.R_off[r], nl[x], s);
			h.n_hits[r] = (int32_t) nl[x]; h.best_score[r] = s;
		}
	}
	kind[j] = kd;
}

}  // namespace

int kmahip_stage3a_pe(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_pe_recs *recs, const kmahip_params *p, kmahip_hits *out, int32_t *pe_kind, hipStream_t stream) {
	if(!g_mem_mode) return kmahip_launch_align_pe(db, ws, reads, recs, p, out, pe_kind, stream);
	if(!db->dev.tlen) { kmahip_set_error("index has no .length.b"); return KMAHIP_EINVAL; }
	const int64_t np = reads->n_reads / 2;
	if(np) hipLaunchKernelGGL(mem_hits_pe_kernel, dim3((unsigned) ((np + 255) / 256)), dim3(256), 0, stream, np, (int) db->info.kmersize, reads->len, *recs, db->dev.tlen, *out, pe_kind,
	                          (unsigned long long *) out->alignment_scores, (unsigned long long *) out->uniq_alignment_scores);
	HIP_TRY(hipGetLastError());
	return KMAHIP_OK;
}

int kmahip_stage3a_se(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_cands *cands, const kmahip_params *p, kmahip_hits *out, hipStream_t stream) {
	if(!g_mem_mode) return kmahip_launch_align_se(db, ws, reads, cands, p, out, stream);
	if(!db->dev.tlen) { kmahip_set_error("index has no .length.b"); return KMAHIP_EINVAL; }
	const int64_t n = reads->n_reads;
	// (a record with a negative score whose list ends on a forward template would be filed with a negative count and take anker_rc in
	// stage 3c, runkma.c:1124: only forced pairing writes such a list, and the whole runs refuse it -- the single-end finders and the
	// penalty / union pairing put the reverse strand's templates last)
	if(n) hipLaunchKernelGGL(mem_hits_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, n, (int) db->info.kmersize, reads->len, cands->rc_flag, cands->flag, cands->T_off,
	                         cands->T, db->dev.tlen, *out, (unsigned long long *) out->alignment_scores, (unsigned long long *) out->uniq_alignment_scores);
	HIP_TRY(hipGetLastError());
	return KMAHIP_OK;
}

// everything behind stage 2 on a batch that is in HBM with its candidate lists: stage 3a, ConClave + the `.res` statistics, the
// traceback, the pile-up. per_read: host arrays for the columns a `.frag` writer needs (any may be NULL).
struct PerRead { int32_t *tmpl, *n_hits, *rc, *trace_stats; };

static int run_after_stage2(kmahip_db *db, kmahip_ws *ws, DevBlock &B, const kmahip_reads &d, kmahip_cands &c, int64_t total, const kmahip_params *p,
                            double evalue, int bcd, int64_t max_frag, kmahip_run *out, const PerRead &per_read, std::chrono::steady_clock::time_point &t) {
	const int64_t n = d.n_reads;
	const size_t D = db->info.DB_size;
	hipStream_t s = 0;
	const bool dbg = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
	int rc;
	// stage 3a
	kmahip_hits h;
	if((rc = B.get((size_t) n + 1, &h.n_hits, true)) || (rc = B.get((size_t) n + 1, &h.best_score, true)) || (rc = B.get((size_t) n + 1, &h.flag, true)) ||
	   (rc = B.get((size_t) n + 1, &h.rc, true)) || (rc = B.get((size_t) total + 1, &h.tmpl, true)) || (rc = B.get((size_t) total + 1, &h.score, true)) ||
	   (rc = B.get((size_t) total + 1, &h.start, true)) || (rc = B.get((size_t) total + 1, &h.end, true)) ||
	   (rc = B.get(D, &h.alignment_scores, true)) || (rc = B.get(D, &h.uniq_alignment_scores, true))) return rc;
	if(dbg) { auto t2 = t; fprintf(stderr, "[kmahip] run_se: stage 3a buffers after %.2f ms\n", since(t2)); }
	for(;;) {
		if(n && (rc = kmahip_stage3a_se(db, ws, &d, &c, p, &h, s))) return rc;
		if(dbg) { auto t2 = t; fprintf(stderr, "[kmahip] run_se: stage 3a launched after %.2f ms\n", since(t2)); }
		HIP_TRY(hipStreamSynchronize(s));
		if(ws_status(ws, nullptr) != 3) break;
		if(!grow_mem_cap(ws)) { kmahip_set_error("seed (MEM) capacity per read/template pair exceeded"); return KMAHIP_EOVERFLOW; }
		HIP_TRY(hipMemsetAsync(h.alignment_scores, 0, D * 8, s)); HIP_TRY(hipMemsetAsync(h.uniq_alignment_scores, 0, D * 8, s));
	}
	out->ms[1] = since(t);

	// stage 3b + the `.res` statistics (host arithmetic on one u64 per template)
	kmahip_conclave cc;
	if((rc = B.get((size_t) n + 1, &cc.tmpl, true)) || (rc = B.get((size_t) n + 1, &cc.start, true)) || (rc = B.get((size_t) n + 1, &cc.end, true)) ||
	   (rc = B.get(D, &cc.w_scores, true))) return rc;
	cc.fragment_counts = nullptr; cc.read_counts = nullptr; cc.depth = nullptr;
	if(dbg) { auto t2 = t; fprintf(stderr, "[kmahip] run_se: ConClave buffers after %.2f ms\n", since(t2)); }
	if(n && (rc = kmahip_conclave_se_dev(db, ws, &d, &c, &h, &cc, s))) return rc;
	std::vector<uint64_t> w(D);
	HIP_TRY(hipMemcpy(w.data(), cc.w_scores, D * 8, hipMemcpyDeviceToHost));
	if(dbg) { auto t2 = t; fprintf(stderr, "[kmahip] run_se: ConClave kernel + scores back after %.2f ms\n", since(t2)); }
	if((rc = kmahip_res_rows(db, w.data(), evalue, p->scoreT, out->rows, out->rows_cap, &out->n_rows))) return rc;
	if(dbg) { auto t2 = t; fprintf(stderr, "[kmahip] run_se: .res statistics after %.2f ms\n", since(t2)); }
	std::vector<uint8_t> ok(D + 8, 0);
	for(int64_t r = 0; r < out->n_rows; ++r) ok[(size_t) out->rows[r].template_id] = (uint8_t) out->rows[r].significant;
	const uint8_t *d_ok = nullptr;
	if((rc = B.up(ok.data(), D + 8, 0, &d_ok))) return rc;
	out->ms[2] = since(t);

	// stage 3c per read; the run pool is sized for a handful of runs per read and grown on demand
	kmahip_traces tr;
	if((rc = B.get((size_t) 10 * n + 10, &tr.stats)) || (rc = B.get((size_t) n + 1, &tr.ops_off)) || (rc = B.get((size_t) n + 1, &tr.n_ops))) return rc;
	tr.ops_cap = 6 * n + (1 << 20);
	for(int attempt = 0; n; ++attempt) {
		if((rc = B.get((size_t) tr.ops_cap, &tr.ops))) return rc;
		if((rc = kmahip_launch_trace(db, ws, &d, h.rc, cc.tmpl, d_ok, p, &tr, s))) return rc;
		HIP_TRY(hipStreamSynchronize(s));
		unsigned long long used = 0;
		const int st = ws_status(ws, &used);
		if(st == 2 || (int64_t) used > tr.ops_cap) {
			if(attempt >= 2) { kmahip_set_error("alignment run pool: %llu runs needed", used); return KMAHIP_EOVERFLOW; }
			tr.ops_cap = (int64_t) used + (1 << 20);
			continue;
		}
		if(st == 16 && grow_mem_cap(ws)) { --attempt; continue; }
		if(st) { kmahip_set_error("trace stage: a read needs more scratch than the workspace holds (status %d)", st); return KMAHIP_EDEVICE; }
		break;
	}
	out->ms[3] = since(t);

	// stage 3c per template
	if(n) {
		kmahip_assemble_opts ao = {max_frag, evalue, bcd, 0, out->caller, out->sig90, nullptr, out->support};      // (caller / sig90: `-bcNano`)
		if((rc = kmahip_assemble2_dev(db, ws, &d, h.rc, cc.tmpl, &tr, &ao, &out->assembly))) return rc;
	} else for(size_t i = 0; i < D; ++i) { out->assembly.cover[i] = 0; out->assembly.aln_len[i] = 0; out->assembly.depth[i] = 0; out->assembly.asm_len[i] = 0; }
	out->ms[4] = since(t);     // (pile-up + copy-back + consensus: kmahip_assemble_dev prints the split with KMAHIP_DEBUG_TIMING)

	// per-read columns for the text writers
	if(n) {
		if(per_read.tmpl) HIP_TRY(hipMemcpy(per_read.tmpl, cc.tmpl, (size_t) n * 4, hipMemcpyDeviceToHost));
		if(per_read.n_hits) HIP_TRY(hipMemcpy(per_read.n_hits, h.n_hits, (size_t) n * 4, hipMemcpyDeviceToHost));
		if(per_read.rc) HIP_TRY(hipMemcpy(per_read.rc, h.rc, (size_t) n * 4, hipMemcpyDeviceToHost));
		if(per_read.trace_stats) HIP_TRY(hipMemcpy(per_read.trace_stats, tr.stats, (size_t) n * 40, hipMemcpyDeviceToHost));
	}
	out->ms[5] = since(t);
	return KMAHIP_OK;
}

extern "C" int kmahip_run_se(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p, double evalue, int bcd,
                             int64_t max_frag, kmahip_run *out) {
	if(!db || !ws || !reads || !p || !out || !out->rows || !out->assembly.cover || !out->assembly.aln_len || !out->assembly.depth || !out->assembly.asm_len) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	const int64_t n = reads->n_reads;
	if(n < 0 || reads->seq_words < 0 || reads->N_total < 0) { kmahip_set_error("negative size"); return KMAHIP_EINVAL; }
	for(int i = 0; i < 6; ++i) out->ms[i] = 0;
	out->n_rows = 0;
	hipStream_t s = 0;
	const bool dbg = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
	DevBlock B;
	int rc;
	auto t = std::chrono::steady_clock::now();

	// the batch, once (the slabs are sized for what a run of n reads usually needs: ~230 bytes per read next to the reads)
	B.expect((size_t) reads->seq_words * 8 + (size_t) reads->N_total * 4 + (size_t) n * 280 + (64u << 20));
	kmahip_reads d = *reads;
	d.q_start = nullptr; d.q_end = nullptr;       // (host pointers, if any: this entry point maps whole reads)
	if((rc = B.up(reads->seq, (size_t) reads->seq_words, 2, &d.seq)) || (rc = B.up(reads->seq_off, (size_t) n + 1, 0, &d.seq_off)) ||
	   (rc = B.up(reads->len, (size_t) n, 1, &d.len)) || (rc = B.up(reads->N, (size_t) reads->N_total, 1, &d.N)) ||
	   (rc = B.up(reads->N_off, (size_t) n + 1, 0, &d.N_off))) return rc;
	HIP_TRY(hipStreamSynchronize(s));
	out->ms[0] = since(t);

	// stage 2; the candidate lists have no bound known in advance: start at 2 per read and redo with the exact size if short
	kmahip_cands c;
	if((rc = B.get((size_t) n + 1, &c.rc_flag)) || (rc = B.get((size_t) n + 1, &c.flag)) || (rc = B.get((size_t) n + 1, &c.T_off))) return rc;
	int64_t total = 0;
	c.T_cap = 2 * n + 4096;
	for(int attempt = 0;; ++attempt) {
		if((rc = B.get((size_t) c.T_cap, &c.T))) return rc;
		if((rc = kmahip_launch_scan_se(db, ws, &d, p, &c, s))) return rc;
		HIP_TRY(hipStreamSynchronize(s));
		const int st = ws_status(ws, nullptr);
		if(st == 1) {
			if(attempt >= 4) { kmahip_set_error("internal candidate pool exhausted"); return KMAHIP_EOVERFLOW; }
			ws->pool_scale *= 2; ws->cap_reads = 0;     // grown by the next launch
			continue;
		}
		HIP_TRY(hipMemcpy(&total, c.T_off + n, sizeof total, hipMemcpyDeviceToHost));
		if(dbg) { auto t2 = t; fprintf(stderr, "[kmahip] run_se: scan attempt %d: %.1f ms, %lld candidates (cap %lld)\n", attempt, since(t2), (long long) total, (long long) c.T_cap); }
		if(total <= c.T_cap) break;
		c.T_cap = total + 1024;
	}
	if(dbg) { auto t2 = t; fprintf(stderr, "[kmahip] run_se: stage 2 done after %.1f ms\n", since(t2)); }

	PerRead pr = {out->tmpl, out->n_hits, out->rc, out->trace_stats};
	return run_after_stage2(db, ws, B, d, c, total, p, evalue, bcd, max_frag, out, pr, t);
}

// ---- paired run (`-ipe r1 r2 -apm p -1t1`): the batch goes up once; stages 2 and 3a of the pairs and of the reads filed singly, the
// merge of their frag_raw records in stream order, ConClave, the fragments in record order, the traceback and the pile-up all work
// on what is in HBM (runKMA + save_kmers_pair + alnFragsPE + runConClave + assemble_KMA for one chunk of input) ---------------------
int kmahip_frag_write_src(const char *path, kmahip_db *db, const kmahip_reads *reads, int64_t n, const int64_t *src, const int32_t *rc,
                          const int32_t *tmpl, const int32_t *n_hits, const int32_t *trace_stats, int stats_stride, int64_t max_frag, int order,
                          const int64_t *frag_rank, const char *read_names, const int64_t *read_name_off, int64_t *rows);      // fragout.hip

namespace {

// reads idx[0 .. m) of a batch in HBM as a batch of its own (each read followed by one pad word, like the source)
__global__ __launch_bounds__(256) void gather_sizes_kernel(int64_t m, const int64_t *idx, const int32_t *len, const int64_t *N_off, int64_t *words, int64_t *n_N) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i > m) return;
	if(i == m) { words[i] = 0; n_N[i] = 0; return; }
	const int64_t r = idx[i];
	words[i] = ((len[r] + 31) >> 5) + 1;
	n_N[i] = N_off[r + 1] - N_off[r];
}

__global__ __launch_bounds__(256) void gather_copy_kernel(int64_t m, const int64_t *idx, const uint64_t *seq, const int64_t *seq_off, const int32_t *len,
                                                          const int32_t *N, const int64_t *N_off, uint64_t *o_seq, const int64_t *o_seq_off, int32_t *o_len,
                                                          int32_t *o_N, const int64_t *o_N_off) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= m) return;
	const int64_t r = idx[i];
	const int L = len[r], w = (L + 31) >> 5;
	const uint64_t *a = seq + seq_off[r];
	uint64_t *b = o_seq + o_seq_off[i];
	for(int x = 0; x < w; ++x) b[x] = a[x];
	b[w] = 0;
	o_len[i] = L;
	const int32_t *na = N + N_off[r];
	int32_t *nb = o_N + o_N_off[i];
	const int nn = (int) (N_off[r + 1] - N_off[r]);
	for(int x = 0; x < nn; ++x) nb[x] = na[x];
}

int gather_batch(DevBlock &B, const kmahip_reads &src, const int64_t *d_idx, int64_t m, kmahip_reads *dst, hipStream_t s) {
	int64_t *wc, *nc, *so, *no;
	int rc;
	if((rc = B.get((size_t) m + 1, &wc)) || (rc = B.get((size_t) m + 1, &nc)) || (rc = B.get((size_t) m + 1, &so)) || (rc = B.get((size_t) m + 1, &no))) return rc;
	hipLaunchKernelGGL(gather_sizes_kernel, dim3((unsigned) ((m + 256) / 256)), dim3(256), 0, s, m, d_idx, src.len, src.N_off, wc, nc);
	if((rc = scan_i64(B, wc, so, (size_t) m + 1, s)) || (rc = scan_i64(B, nc, no, (size_t) m + 1, s))) return rc;
	int64_t tw = 0, tn = 0;
	HIP_TRY(hipMemcpyAsync(&tw, so + m, 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(&tn, no + m, 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	uint64_t *seq;
	int32_t *len, *N;
	if((rc = B.get((size_t) tw + 2, &seq)) || (rc = B.get((size_t) m + 1, &len)) || (rc = B.get((size_t) tn + 1, &N))) return rc;
	HIP_TRY(hipMemsetAsync(seq + tw, 0, 16, s));
	HIP_TRY(hipMemsetAsync(len + m, 0, 4, s));
	if(m) hipLaunchKernelGGL(gather_copy_kernel, dim3((unsigned) ((m + 255) / 256)), dim3(256), 0, s, m, d_idx, src.seq, src.seq_off, src.len, src.N, src.N_off,
	                         seq, so, len, N, no);
	HIP_TRY(hipGetLastError());
	*dst = kmahip_reads{};
	dst->n_reads = m; dst->seq = seq; dst->seq_off = so; dst->len = len; dst->N = N; dst->N_off = no;
	dst->seq_words = tw; dst->N_total = tn; dst->max_len = src.max_len;
	return KMAHIP_OK;
}

// the same over two batches in HBM: idx < n0 is a read of `a`, idx >= n0 read idx - n0 of `b` (the paired run of the default mode: the
// couples' reads are the uploaded batch's, a singly loaded read's fragments are the chain finder's records, which carry their query
// bounds -- a read of `a` gets the whole read as its bounds)
struct Gather2 {
	const uint64_t *seq[2];
	const int64_t *seq_off[2], *N_off[2];
	const int32_t *len[2], *N[2];
	const int32_t *qs, *qe;          // of `b`
	int64_t n0;
};

__global__ __launch_bounds__(256) void gather_sizes2_kernel(int64_t m, const int64_t *idx, const Gather2 G, int64_t *words, int64_t *n_N) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i > m) return;
	if(i == m) { words[i] = 0; n_N[i] = 0; return; }
	const int k = idx[i] >= G.n0;
	const int64_t r = idx[i] - (k ? G.n0 : 0);
	words[i] = ((G.len[k][r] + 31) >> 5) + 1;
	n_N[i] = G.N_off[k][r + 1] - G.N_off[k][r];
}

__global__ __launch_bounds__(256) void gather_copy2_kernel(int64_t m, const int64_t *idx, const Gather2 G, uint64_t *o_seq, const int64_t *o_seq_off, int32_t *o_len,
                                                           int32_t *o_N, const int64_t *o_N_off, int32_t *o_qs, int32_t *o_qe) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= m) return;
	const int k = idx[i] >= G.n0;
	const int64_t r = idx[i] - (k ? G.n0 : 0);
	const int L = G.len[k][r], w = (L + 31) >> 5;
	const uint64_t *a = G.seq[k] + G.seq_off[k][r];
	uint64_t *b = o_seq + o_seq_off[i];
	for(int x = 0; x < w; ++x) b[x] = a[x];
	b[w] = 0;
	o_len[i] = L;
	const int32_t *na = G.N[k] + G.N_off[k][r];
	int32_t *nb = o_N + o_N_off[i];
	const int nn = (int) (G.N_off[k][r + 1] - G.N_off[k][r]);
	for(int x = 0; x < nn; ++x) nb[x] = na[x];
	o_qs[i] = k ? G.qs[r] : 0; o_qe[i] = k ? G.qe[r] : L;
}

int gather_batch2(DevBlock &B, const kmahip_reads &a, int64_t n0, const kmahip_reads &b, const int64_t *d_idx, int64_t m, kmahip_reads *dst, hipStream_t s) {
	int64_t *wc, *nc, *so, *no;
	int rc;
	Gather2 G;
	G.seq[0] = a.seq; G.seq_off[0] = a.seq_off; G.N_off[0] = a.N_off; G.len[0] = a.len; G.N[0] = a.N;
	G.seq[1] = b.seq; G.seq_off[1] = b.seq_off; G.N_off[1] = b.N_off; G.len[1] = b.len; G.N[1] = b.N;
	G.qs = b.q_start; G.qe = b.q_end; G.n0 = n0;
	if((rc = B.get((size_t) m + 1, &wc)) || (rc = B.get((size_t) m + 1, &nc)) || (rc = B.get((size_t) m + 1, &so)) || (rc = B.get((size_t) m + 1, &no))) return rc;
	hipLaunchKernelGGL(gather_sizes2_kernel, dim3((unsigned) ((m + 256) / 256)), dim3(256), 0, s, m, d_idx, G, wc, nc);
	if((rc = scan_i64(B, wc, so, (size_t) m + 1, s)) || (rc = scan_i64(B, nc, no, (size_t) m + 1, s))) return rc;
	int64_t tw = 0, tn = 0;
	HIP_TRY(hipMemcpyAsync(&tw, so + m, 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipMemcpyAsync(&tn, no + m, 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	uint64_t *seq;
	int32_t *len, *N, *qs, *qe;
	if((rc = B.get((size_t) tw + 2, &seq)) || (rc = B.get((size_t) m + 1, &len)) || (rc = B.get((size_t) tn + 1, &N)) || (rc = B.get((size_t) m + 1, &qs)) ||
	   (rc = B.get((size_t) m + 1, &qe))) return rc;
	HIP_TRY(hipMemsetAsync(seq + tw, 0, 16, s));
	HIP_TRY(hipMemsetAsync(len + m, 0, 4, s));
	if(m) hipLaunchKernelGGL(gather_copy2_kernel, dim3((unsigned) ((m + 255) / 256)), dim3(256), 0, s, m, d_idx, G, seq, so, len, N, no, qs, qe);
	HIP_TRY(hipGetLastError());
	*dst = kmahip_reads{};
	dst->n_reads = m; dst->seq = seq; dst->seq_off = so; dst->len = len; dst->N = N; dst->N_off = no;
	dst->seq_words = tw; dst->N_total = tn; dst->max_len = a.max_len > b.max_len ? a.max_len : b.max_len;
	dst->q_start = qs; dst->q_end = qe;
	return KMAHIP_OK;
}

// where a fragment's header is: the read itself, or the read a chain record was cut from
__global__ __launch_bounds__(256) void pe_name_src_kernel(int64_t nf, const int64_t *f_src, int64_t n0, const int64_t *rec_read, int64_t *f_name) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i < nf) f_name[i] = f_src[i] < n0 ? f_src[i] : rec_read[f_src[i] - n0];
}

// The frag_raw records of the stream (update_Scores_pe / _se, updatescores.c:300-488), two slots per unit in stream order: a slot
// left at n = 0, score = 0 is no record. A record's hit list lies at r_off of the joint hit arrays (pairs first, the singles'
// lists from s_base on); its one or two fragments are reads of the uploaded batch.
struct RecArgs {
	int64_t n_units;
	const int32_t *u_first, *u_idx;      // first read of the unit; >= 0: pair number, < 0: -(single number) - 1
	const int32_t *p_len, *s_len;        // lengths in the pair batch (mates interleaved) / the single batch
	const int32_t *mate, *p_n, *p_best, *p_rc, *kind;
	const int64_t *R_off;
	const int32_t *s_n, *s_best, *s_rc;
	const int64_t *T_off;
	int64_t s_base;
	int32_t *r_n, *r_score, *r_ql, *r_ql2;
	int64_t *r_off;
	int32_t *fr_read, *fr_rc;            // 2 per slot (-1: none)
};

__global__ __launch_bounds__(256) void pe_records_kernel(const RecArgs A) {
	const int64_t u = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(u >= A.n_units) return;
	const int64_t s0 = 2 * u;
	for(int x = 0; x < 2; ++x) {
		A.r_n[s0 + x] = 0; A.r_score[s0 + x] = 0; A.r_ql[s0 + x] = 0; A.r_ql2[s0 + x] = 0; A.r_off[s0 + x] = 0;
		A.fr_read[2 * (s0 + x)] = -1; A.fr_read[2 * (s0 + x) + 1] = -1; A.fr_rc[2 * (s0 + x)] = 0; A.fr_rc[2 * (s0 + x) + 1] = 0;
	}
	auto rec = [&](int64_t slot, int n, int score, int ql, int ql2, int64_t off) {
		A.r_n[slot] = n; A.r_score[slot] = score; A.r_ql[slot] = ql; A.r_ql2[slot] = ql2; A.r_off[slot] = off;
	};
	const int first = A.u_first[u];
	if(A.u_idx[u] < 0) {
		const int64_t j = -(int64_t) A.u_idx[u] - 1;
		if(A.s_n[j] > 0) {
			rec(s0, A.s_n[j], A.s_best[j], A.s_len[j], 0, A.s_base + A.T_off[j]);
			A.fr_read[2 * s0] = first; A.fr_rc[2 * s0] = A.s_rc[j];
		}
		return;
	}
	const int64_t j = A.u_idx[u], r0 = 2 * j, r1 = 2 * j + 1;
	auto ln = [&](int64_t x) { return A.p_len[2 * j + A.mate[x]]; };
	auto frag = [&](int64_t slot, int at, int64_t x) { A.fr_read[2 * slot + at] = first + A.mate[x]; A.fr_rc[2 * slot + at] = A.p_rc[x] & 1; };
	const int64_t o = A.R_off[r1];
	const int kd = A.kind[j];
	if(kd == 1) {
		const bool swapped = (A.p_rc[r1] & 2) != 0;      // the second slot's fragment is written first (alnfrags.c:1807-1812)
		rec(s0, A.p_n[r1], -A.p_best[r1], swapped ? ln(r1) : ln(r0), swapped ? ln(r0) : ln(r1), o);
		frag(s0, 0, swapped ? r1 : r0); frag(s0, 1, swapped ? r0 : r1);
	} else if(kd == 2) {
		const int n0 = A.p_n[r0];
		rec(s0, n0, A.p_best[r0], ln(r0), 0, o); frag(s0, 0, r0);
		rec(s0 + 1, A.p_n[r1], A.p_best[r1], ln(r1), 0, o + n0); frag(s0 + 1, 0, r1);
	} else if(kd == 3 || kd == 4) {
		const int64_t x = kd == 3 ? r0 : r1;
		rec(s0, A.p_n[x], A.p_best[x], ln(x), 0, o); frag(s0, 0, x);
	} else {
		for(int64_t x = r0; x <= r1; ++x) if(A.mate[x] >= 0 && A.p_n[x] > 0) { rec(s0 + (x - r0), A.p_n[x], A.p_best[x], ln(x), 0, A.R_off[x]); frag(s0 + (x - r0), 0, x); }
	}
}

// fragments that ConClave filed (template != 0), per slot
__global__ __launch_bounds__(256) void pe_frag_count_kernel(int64_t n_slots, const int32_t *c_tmpl, const int32_t *fr_read, int64_t *cnt, uint8_t *cnt8) {
	const int64_t s = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(s > n_slots) return;
	int c = 0;
	if(s < n_slots && c_tmpl[s] != 0) c = (fr_read[2 * s] >= 0) + (fr_read[2 * s + 1] >= 0);
	cnt[s] = c;
	if(s < n_slots) cnt8[s] = (uint8_t) c;
}

// The filed fragments in record order. The first fragment of a record carries the sign of the template (conclave.c:131-146).
// runConClave closes a chunk of filed fragments when, AFTER a whole record, maxFrag or more have gone in (conclave.c:164-196):
// a couple that straddles the limit makes a chunk of maxFrag + 1. starts[c] = filed fragments before chunk c (counted record by
// record on the host); a fragment is handed on as position c (maxFrag + 1) + index, with maxFrag + 1 as the chunk length the
// pile-up and the writer divide by.
__global__ __launch_bounds__(256) void pe_frag_fill_kernel(int64_t n_slots, const int32_t *c_tmpl, const int32_t *r_n, const int32_t *fr_read, const int32_t *fr_rc,
                                                           const int64_t *f_off, const int64_t *starts, int n_starts, int64_t mf, int64_t chunk_base, int64_t *f_src, int32_t *f_rc,
                                                           int32_t *f_t, int32_t *f_nh, int64_t *f_rank) {
	const int64_t s = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(s >= n_slots) return;
	const int tt = c_tmpl[s];
	if(tt == 0) return;
	int64_t g = f_off[s];
	for(int x = 0; x < 2; ++x) {
		if(fr_read[2 * s + x] < 0) continue;
		int lo = 0, hi = n_starts;          // last chunk that starts at or before g
		while(hi - lo > 1) { const int mid = (lo + hi) >> 1; if(starts[mid] <= g) lo = mid; else hi = mid; }
		f_src[g] = fr_read[2 * s + x]; f_rc[g] = fr_rc[2 * s + x]; f_t[g] = x == 0 ? tt : abs(tt); f_nh[g] = r_n[s];
		f_rank[g] = (chunk_base + lo) * (mf + 1) + (g - starts[lo]);
		++g;
	}
}

// what a `.frag` row prints of a fragment's statistics: score, start, end, kept
__global__ __launch_bounds__(256) void pe_stats4_kernel(int64_t n, const int32_t *stats, int32_t *out) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n) return;
	for(int x = 0; x < 4; ++x) out[4 * i + x] = stats[10 * i + x];
}

}  // namespace

int kmahip_conclave_records_carry(kmahip_db *db, int64_t n_records, const int32_t *q_len, const int32_t *q_len2, const int64_t *off, const kmahip_hits *hits,
                                  kmahip_conclave *out, const int32_t carry[3], hipStream_t stream);                                    // conclave.hip
int kmahip_conclave_records_last(kmahip_db *db, int64_t n_records, const int32_t *q_len, const int32_t *q_len2, const int64_t *off, const kmahip_hits *hits,
                                 int32_t *d_last, hipStream_t stream);
// what a read-sharded run adds to the paired run: the communicator and where the owners' results go
struct ShardCtx {
	kmahip_comm *comm;
	const kmahip_shard_opts *opts;
	const char *out_prefix;
	double *ms;
	std::vector<uint64_t> frag_counts;       // summed over the ranks: the owners' template ranges are cut by them
};
static int shard_pe_tail(ShardCtx *sc, kmahip_db *db, kmahip_ws *ws, DevBlock &B, const kmahip_read_batch *batch, const kmahip_reads &dF, const int32_t *f_rc, const int32_t *f_t,
                         const int32_t *f_nh, const int64_t *f_rank, const kmahip_traces &tr, const int64_t *h_src, const kmahip_res_row *rows, int64_t n_rows, int64_t chunk,
                         std::chrono::steady_clock::time_point &t);
static int shard_allreduce(ShardCtx *sc, uint64_t *d_buf, size_t n);
static int shard_carry_in(ShardCtx *sc, const int32_t last[4], int32_t carry[3]);
static int shard_chunk_token(ShardCtx *sc, bool receive, int64_t state[2]);
static uint64_t *shard_frag_counts(ShardCtx *sc, size_t D);

// stage 2 of the default mode on an uploaded batch and its records as a batch of their own, in stream order: R.d = the read of a
// record, or its reverse complement where the record prints that, with the record's query bounds; R.c = the template lists
struct ChainRecs {
	int64_t m = 0, n_T = 0;
	kmahip_reads d{};
	kmahip_cands c{};
	int64_t *o_read = nullptr;     // device: the read a record comes from
	int32_t *o_emit = nullptr;     // device: 1 = the record holds the reverse complement
};
static int chain_records(kmahip_db *db, kmahip_ws *ws, DevBlock &B, const kmahip_reads &dR, const kmahip_reads *reads, const kmahip_params *p,
                         const kmahip_chain_params *cp, ChainRecs &R, const std::function<void(const char *)> &lap);
static int run_pe_impl(kmahip_db *db, kmahip_ws *ws, const kmahip_read_batch *batch, const kmahip_params *p, double evalue, int bcd,
                       int64_t max_frag, const char *frag_path, kmahip_run *out, ShardCtx *sc, const KmaPeDev *pd);

extern "C" int kmahip_run_pe(kmahip_db *db, kmahip_ws *ws, const kmahip_read_batch *batch, const kmahip_params *p, double evalue, int bcd,
                             int64_t max_frag, const char *frag_path, kmahip_run *out) {
	return run_pe_impl(db, ws, batch, p, evalue, bcd, max_frag, frag_path, out, nullptr, nullptr);
}

// the paired run on reads and headers that are in HBM already (the batched session, session.hip: batch->reads holds DEVICE arrays,
// batch->pair the host's flags, the headers are pd's)
int kmahip_run_pe_resident(kmahip_db *db, kmahip_ws *ws, const kmahip_read_batch *batch, const KmaPeDev *pd, const kmahip_params *p, double evalue, int bcd,
                           int64_t max_frag, const char *frag_path, kmahip_run *out) {
	return run_pe_impl(db, ws, batch, p, evalue, bcd, max_frag, frag_path, out, nullptr, pd);
}

static int run_pe_impl(kmahip_db *db, kmahip_ws *ws, const kmahip_read_batch *batch, const kmahip_params *p, double evalue, int bcd,
                       int64_t max_frag, const char *frag_path, kmahip_run *out, ShardCtx *sc, const KmaPeDev *pd) {
	if(!db || !ws || !batch || !p || !out || !out->rows || !out->assembly.cover || !out->assembly.aln_len || !out->assembly.depth || !out->assembly.asm_len ||
	   (!batch->pair && batch->reads.n_reads > 0)) {
		kmahip_set_error("null argument"); return KMAHIP_EINVAL;
	}
	const kmahip_reads &R = batch->reads;
	const int64_t n = R.n_reads;
	if(n < 0 || n > 0x7ffffff0ll || R.seq_words < 0 || R.N_total < 0) { kmahip_set_error("bad batch size"); return KMAHIP_EINVAL; }
	if(frag_path && n && !pd && (!batch->names || !batch->name_off)) { kmahip_set_error("the batch carries no read names"); return KMAHIP_EINVAL; }
	if(pd && sc) { kmahip_set_error("a resident batch is not sharded"); return KMAHIP_EINVAL; }
	const size_t D = db->info.DB_size;
	for(int i = 0; i < 6; ++i) out->ms[i] = 0;
	out->n_rows = 0;
	hipStream_t s = 0;
	const bool dbg = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
	auto t = std::chrono::steady_clock::now();
	auto stamp = [&](const char *what) { if(dbg) { auto t2 = t; fprintf(stderr, "[kmahip] run_pe: %s after %.2f ms\n", what, since(t2)); } };
	int rc;

	// units of the stream: a pair (two reads) or a single
	std::unique_ptr<int32_t[]> u_first(new int32_t[(size_t) n + 1]), u_idx(new int32_t[(size_t) n + 1]);
	// the default mode (no -1t1): a singly loaded read goes to the chain finder, couples to the pairing as ever (savekmers.c:196-200)
	const kmahip_chain_params *cp = ws->pe_chain_on ? &ws->pe_chain : nullptr;
	int64_t U = 0, np = 0, ns = 0;
	for(int64_t i = 0; i < n; ++U) {
		u_first[(size_t) U] = (int32_t) i;
		if(batch->pair[i] == 1 && i + 1 < n && batch->pair[i + 1] == 2) { u_idx[(size_t) U] = (int32_t) np++; i += 2; }
		else { u_idx[(size_t) U] = -(int32_t) (ns++) - 1; i += 1; }
	}

	// the batch, once
	DevBlock B;
	// the three pinned text buffers of the fragment writer: pinning 192 MB takes 30 to 150 ms -- on a thread of its own, beside the stages
	struct Pinned {
		char *buf[3] = {nullptr, nullptr, nullptr};
		std::thread th;
		bool ok = false;
		~Pinned() { if(th.joinable()) th.join(); for(int x = 0; x < 3; ++x) if(buf[x]) (void) hipHostFree(buf[x]); }
	} pinned;
	const int64_t pin_chunk = 64ll << 20;
	if(frag_path && !sc && !pd && n > 100000 && !getenv("KMAHIP_PE_HOST_FRAG")) {
		int dev = 0;
		(void) hipGetDevice(&dev);
		pinned.th = std::thread([&pinned, dev, pin_chunk] {
			(void) hipSetDevice(dev);
			bool ok = true;
			for(int x = 0; x < 3 && ok; ++x) ok = hipHostMalloc((void **) &pinned.buf[x], (size_t) pin_chunk + 16, hipHostMallocDefault) == hipSuccess;
			pinned.ok = ok;
		});
	}
	B.expect((size_t) R.seq_words * 16 + (size_t) R.N_total * 8 + (size_t) n * 420 + (64u << 20));
	kmahip_reads dR = R;
	dR.q_start = nullptr; dR.q_end = nullptr;
	if(!pd && ((rc = B.up(R.seq, (size_t) R.seq_words, 2, &dR.seq)) || (rc = B.up(R.seq_off, (size_t) n + 1, 0, &dR.seq_off)) ||
	   (rc = B.up(R.len, (size_t) n, 1, &dR.len)) || (rc = B.up(R.N, (size_t) R.N_total, 1, &dR.N)) || (rc = B.up(R.N_off, (size_t) n + 1, 0, &dR.N_off)))) return rc;
	kmahip_reads dP = dR, dS = dR;
	dS.n_reads = 0;
	std::vector<int64_t> s_idx;
	if(ns > 0) {          // pairs and singles as batches of their own (all pairs: the batch as it is)
		std::vector<int64_t> p_idx;
		p_idx.reserve((size_t) 2 * np); s_idx.reserve((size_t) ns);
		for(int64_t u = 0; u < U; ++u) {
			if(u_idx[(size_t) u] < 0) s_idx.push_back(u_first[(size_t) u]);
			else { p_idx.push_back(u_first[(size_t) u]); p_idx.push_back(u_first[(size_t) u] + 1); }
		}
		const int64_t *d_pi = nullptr, *d_si = nullptr;
		if((rc = B.up(p_idx.data(), (size_t) 2 * np, 1, &d_pi)) || (rc = B.up(s_idx.data(), (size_t) ns, 1, &d_si))) return rc;
		if((rc = gather_batch(B, dR, d_pi, 2 * np, &dP, s)) || (rc = gather_batch(B, dR, d_si, ns, &dS, s))) return rc;
	}
	HIP_TRY(hipStreamSynchronize(s));
	out->ms[0] = since(t);
	// (the columns of the `.frag` rows, at most one fragment per read)
	HostCols HC;
	int64_t *h_src = nullptr, *h_rank = nullptr;
	int32_t *h_rc = nullptr, *h_t = nullptr, *h_nh = nullptr, *h_stats = nullptr;
	if(frag_path && n && !pd) {
		h_src = HC.get<int64_t>((size_t) n); h_rank = HC.get<int64_t>((size_t) n); h_rc = HC.get<int32_t>((size_t) n); h_t = HC.get<int32_t>((size_t) n);
		h_nh = HC.get<int32_t>((size_t) n); h_stats = HC.get<int32_t>((size_t) n * 4);
		if(!h_src || !h_rank || !h_rc || !h_t || !h_nh || !h_stats) { kmahip_set_error("out of host memory"); return KMAHIP_ENOMEM; }
		HC.start();
	}

	// stage 2: pairs, then the singles
	kmahip_pe_recs recs;
	if((rc = B.get((size_t) 2 * np + 2, &recs.mate)) || (rc = B.get((size_t) 2 * np + 2, &recs.rc)) || (rc = B.get((size_t) 2 * np + 2, &recs.rc_flag)) ||
	   (rc = B.get((size_t) 2 * np + 2, &recs.flag)) || (rc = B.get((size_t) 2 * np + 2, &recs.R_off, true))) return rc;
	int64_t totP = 0, totS = 0;
	recs.T_cap = 4 * np + 4096; recs.T = nullptr;
	for(int attempt = 0; np > 0; ++attempt) {
		if((rc = B.get((size_t) recs.T_cap, &recs.T))) return rc;
		if((rc = kmahip_launch_scan_pe(db, ws, &dP, p, &recs, s))) return rc;
		HIP_TRY(hipStreamSynchronize(s));
		if(ws_status(ws, nullptr) == 1) {
			if(attempt >= 4) { kmahip_set_error("internal candidate pool exhausted"); return KMAHIP_EOVERFLOW; }
			ws->pool_scale *= 2; ws->cap_reads = 0;
			continue;
		}
		HIP_TRY(hipMemcpy(&totP, recs.R_off + 2 * np, sizeof totP, hipMemcpyDeviceToHost));
		if(totP <= recs.T_cap) break;
		if(attempt >= 6) { kmahip_set_error("candidate lists keep growing"); return KMAHIP_EOVERFLOW; }
		recs.T_cap = totP + 1024;
	}
	stamp("stage 2 of the pairs");
	kmahip_cands cd;
	if((rc = B.get((size_t) ns + 1, &cd.rc_flag)) || (rc = B.get((size_t) ns + 1, &cd.flag)) || (rc = B.get((size_t) ns + 1, &cd.T_off, true))) return rc;
	cd.T_cap = 2 * ns + 4096; cd.T = nullptr;
	ChainRecs CR;
	const int64_t *d_rec_read = nullptr;          // (default mode) the read of the uploaded batch a record was cut from
	if(cp && ns > 0) {
		// the singles through save_kmers_chain: zero or more records per read, each a unit of the stream where its read stood, with
		// its query bounds; a record's fragment is read n + record of the two-part gather below
		if((rc = chain_records(db, ws, B, dS, &R, p, cp, CR, [&](const char *what) { stamp(what); }))) return rc;
		const int64_t m = CR.m;
		if(n + m > 0x7ffffff0ll) { kmahip_set_error("bad batch size"); return KMAHIP_EINVAL; }
		std::vector<int64_t> h_or((size_t) m + 1), rec_read((size_t) m + 1, 0);
		if(m) HIP_TRY(hipMemcpy(h_or.data(), CR.o_read, (size_t) m * 8, hipMemcpyDeviceToHost));
		std::unique_ptr<int32_t[]> nu_first(new int32_t[(size_t) (np + m) + 1]), nu_idx(new int32_t[(size_t) (np + m) + 1]);
		int64_t U2 = 0, x = 0;
		for(int64_t u = 0; u < U; ++u) {
			if(u_idx[(size_t) u] >= 0) { nu_first[(size_t) U2] = u_first[(size_t) u]; nu_idx[(size_t) U2] = u_idx[(size_t) u]; ++U2; continue; }
			const int64_t j = -(int64_t) u_idx[(size_t) u] - 1;
			for(; x < m && h_or[(size_t) x] == j; ++x, ++U2) {
				nu_first[(size_t) U2] = (int32_t) (n + x); nu_idx[(size_t) U2] = -(int32_t) x - 1;
				rec_read[(size_t) x] = s_idx[(size_t) j];
			}
		}
		if(x != m) { kmahip_set_error("chain records out of stream order"); return KMAHIP_EDEVICE; }
		U = U2; u_first = std::move(nu_first); u_idx = std::move(nu_idx);
		if((rc = B.up(rec_read.data(), (size_t) m + 1, 1, &d_rec_read))) return rc;
		dS = CR.d; dS.n_reads = m; cd = CR.c; ns = m; totS = CR.n_T;
		if(m == 0) dS.max_len = R.max_len;
	}
	const int64_t n_slots = 2 * U;
	const int32_t *d_first = nullptr, *d_uidx = nullptr;
	if((rc = B.up(u_first.get(), (size_t) U + 1, 1, &d_first)) || (rc = B.up(u_idx.get(), (size_t) U + 1, 1, &d_uidx))) return rc;
	for(int attempt = 0; ns > 0 && !cp; ++attempt) {
		if((rc = B.get((size_t) cd.T_cap, &cd.T))) return rc;
		if((rc = kmahip_launch_scan_se(db, ws, &dS, p, &cd, s))) return rc;
		HIP_TRY(hipStreamSynchronize(s));
		if(ws_status(ws, nullptr) == 1) {
			if(attempt >= 4) { kmahip_set_error("internal candidate pool exhausted"); return KMAHIP_EOVERFLOW; }
			ws->pool_scale *= 2; ws->cap_reads = 0;
			continue;
		}
		HIP_TRY(hipMemcpy(&totS, cd.T_off + ns, sizeof totS, hipMemcpyDeviceToHost));
		if(totS <= cd.T_cap) break;
		if(attempt >= 6) { kmahip_set_error("candidate lists keep growing"); return KMAHIP_EOVERFLOW; }
		cd.T_cap = totS + 1024;
	}

	// stage 3a: one set of hit arrays (the pairs' lists first), one pair of ConClave vectors
	const size_t H = (size_t) totP + (size_t) totS + 2;
	const int64_t s_base = totP + 1;
	kmahip_hits ph, sh;
	int32_t *kind, *H_tmpl, *H_score, *H_start, *H_end;
	uint64_t *AS, *US;
	if((rc = B.get((size_t) 2 * np + 2, &ph.n_hits, true)) || (rc = B.get((size_t) 2 * np + 2, &ph.best_score, true)) || (rc = B.get((size_t) 2 * np + 2, &ph.flag, true)) ||
	   (rc = B.get((size_t) 2 * np + 2, &ph.rc, true)) || (rc = B.get((size_t) np + 1, &kind, true)) ||
	   (rc = B.get((size_t) ns + 1, &sh.n_hits, true)) || (rc = B.get((size_t) ns + 1, &sh.best_score, true)) || (rc = B.get((size_t) ns + 1, &sh.flag, true)) ||
	   (rc = B.get((size_t) ns + 1, &sh.rc, true)) ||
	   (rc = B.get(H, &H_tmpl, true)) || (rc = B.get(H, &H_score, true)) || (rc = B.get(H, &H_start, true)) || (rc = B.get(H, &H_end, true)) ||
	   (rc = B.get(D, &AS, true)) || (rc = B.get(D, &US, true))) return rc;
	ph.tmpl = H_tmpl; ph.score = H_score; ph.start = H_start; ph.end = H_end; ph.alignment_scores = AS; ph.uniq_alignment_scores = US;
	sh.tmpl = H_tmpl + s_base; sh.score = H_score + s_base; sh.start = H_start + s_base; sh.end = H_end + s_base; sh.alignment_scores = AS; sh.uniq_alignment_scores = US;
	for(;;) {
		bool again = false;
		if(np > 0) {
			if((rc = kmahip_stage3a_pe(db, ws, &dP, &recs, p, &ph, kind, s))) return rc;
			HIP_TRY(hipStreamSynchronize(s));
			again = ws_status(ws, nullptr) == 3;
		}
		stamp("stage 3a of the pairs");
		if(ns > 0 && !again) {
			if((rc = kmahip_stage3a_se(db, ws, &dS, &cd, p, &sh, s))) return rc;
			HIP_TRY(hipStreamSynchronize(s));
			again = ws_status(ws, nullptr) == 3;
		}
		if(!again) break;
		if(!grow_mem_cap(ws)) { kmahip_set_error("seed (MEM) capacity per read/template pair exceeded"); return KMAHIP_EOVERFLOW; }
		HIP_TRY(hipMemsetAsync(AS, 0, D * 8, s)); HIP_TRY(hipMemsetAsync(US, 0, D * 8, s));          // (both stages add into them)
	}
	out->ms[1] = since(t);
	if(sc && ((rc = shard_allreduce(sc, AS, D)) || (rc = shard_allreduce(sc, US, D)))) return rc;          // exchange 1

	// the records in stream order, stage 3b over them, the `.res` statistics
	RecArgs A{};
	A.n_units = U; A.u_first = d_first; A.u_idx = d_uidx; A.p_len = dP.len; A.s_len = dS.len;
	A.mate = recs.mate; A.p_n = ph.n_hits; A.p_best = ph.best_score; A.p_rc = ph.rc; A.kind = kind; A.R_off = recs.R_off;
	A.s_n = sh.n_hits; A.s_best = sh.best_score; A.s_rc = sh.rc; A.T_off = cd.T_off; A.s_base = s_base;
	if((rc = B.get((size_t) n_slots + 1, &A.r_n)) || (rc = B.get((size_t) n_slots + 1, &A.r_score)) || (rc = B.get((size_t) n_slots + 1, &A.r_ql)) ||
	   (rc = B.get((size_t) n_slots + 1, &A.r_ql2)) || (rc = B.get((size_t) n_slots + 1, &A.r_off)) || (rc = B.get((size_t) 2 * n_slots + 2, &A.fr_read)) ||
	   (rc = B.get((size_t) 2 * n_slots + 2, &A.fr_rc))) return rc;
	kmahip_conclave cc;
	if((rc = B.get((size_t) n_slots + 1, &cc.tmpl, true)) || (rc = B.get((size_t) n_slots + 1, &cc.start, true)) || (rc = B.get((size_t) n_slots + 1, &cc.end, true)) ||
	   (rc = B.get(D, &cc.w_scores, true))) return rc;
	cc.fragment_counts = nullptr; cc.read_counts = nullptr; cc.depth = nullptr;
	if(sc && (rc = B.get(D, &cc.fragment_counts, true))) return rc;
	{
		if(U > 0) hipLaunchKernelGGL(pe_records_kernel, dim3((unsigned) ((U + 255) / 256)), dim3(256), 0, s, A);
		HIP_TRY(hipGetLastError());
		kmahip_hits h{};
		h.n_hits = A.r_n; h.best_score = A.r_score; h.tmpl = H_tmpl; h.start = H_start; h.end = H_end; h.alignment_scores = AS; h.uniq_alignment_scores = US;
		if(sc) {
			// a record with an empty list takes the first listed hit of the last record before it that had one -- for the first such
			// records of this shard that is a record of an earlier shard (conclave.c:123-127)
			int32_t *d_last = nullptr, last[4] = {0, 0, 0, 0}, carry[3] = {0, 0, 0};
			if((rc = B.get(4, &d_last, true))) return rc;
			if(U > 0 && (rc = kmahip_conclave_records_last(db, n_slots, A.r_ql, A.r_ql2, A.r_off, &h, d_last, s))) return rc;
			HIP_TRY(hipMemcpy(last, d_last, sizeof last, hipMemcpyDeviceToHost));
			if((rc = shard_carry_in(sc, last, carry))) return rc;
			if(U > 0 && (rc = kmahip_conclave_records_carry(db, n_slots, A.r_ql, A.r_ql2, A.r_off, &h, &cc, carry, s))) return rc;
		} else if(U > 0 && (rc = kmahip_conclave_records_dev(db, ws, n_slots, A.r_ql, A.r_ql2, A.r_off, &h, &cc, s))) return rc;
	}
	if(sc) {          // exchange 2: ConClave's per-template outputs summed (the fragment counts cut the owners' template ranges)
		uint64_t *fc = shard_frag_counts(sc, D), *d_fc = nullptr;
		std::vector<uint32_t> c32(D);
		HIP_TRY(hipMemcpy(c32.data(), cc.fragment_counts, D * 4, hipMemcpyDeviceToHost));
		for(size_t x = 0; x < D; ++x) fc[x] = c32[x];
		if((rc = B.get(D, &d_fc))) return rc;
		HIP_TRY(hipMemcpy(d_fc, fc, D * 8, hipMemcpyHostToDevice));
		if((rc = shard_allreduce(sc, (uint64_t *) cc.w_scores, D)) || (rc = shard_allreduce(sc, d_fc, D))) return rc;
		HIP_TRY(hipMemcpy(fc, d_fc, D * 8, hipMemcpyDeviceToHost));
	}
	std::vector<uint64_t> w(D);
	HIP_TRY(hipMemcpy(w.data(), cc.w_scores, D * 8, hipMemcpyDeviceToHost));
	if((rc = kmahip_res_rows(db, w.data(), evalue, p->scoreT, out->rows, out->rows_cap, &out->n_rows))) return rc;
	std::vector<uint8_t> ok(D + 8, 0);
	for(int64_t r = 0; r < out->n_rows; ++r) ok[(size_t) out->rows[r].template_id] = (uint8_t) out->rows[r].significant;
	const uint8_t *d_ok = nullptr;
	if((rc = B.up(ok.data(), D + 8, 0, &d_ok))) return rc;
	out->ms[2] = since(t);

	// the filed fragments in record order as a batch of their own
	const int64_t mf = max_frag > 0 ? max_frag : 1000000;
	int64_t *f_cnt, *f_off;
	uint8_t *d_cnt8;
	if((rc = B.get((size_t) n_slots + 1, &f_cnt)) || (rc = B.get((size_t) n_slots + 1, &f_off)) || (rc = B.get((size_t) n_slots + 1, &d_cnt8))) return rc;
	hipLaunchKernelGGL(pe_frag_count_kernel, dim3((unsigned) ((n_slots + 256) / 256)), dim3(256), 0, s, n_slots, cc.tmpl, A.fr_read, f_cnt, d_cnt8);
	HIP_TRY(hipGetLastError());
	if((rc = scan_i64(B, f_cnt, f_off, (size_t) n_slots + 1, s))) return rc;
	int64_t nf = 0;
	std::vector<uint8_t> cnt8((size_t) n_slots + 1);
	HIP_TRY(hipMemcpyAsync(&nf, f_off + n_slots, 8, hipMemcpyDeviceToHost, s));
	if(n_slots) HIP_TRY(hipMemcpyAsync(cnt8.data(), d_cnt8, (size_t) n_slots, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	// (a read shard continues the chunk the shards before it left open: it starts `in_chunk` fragments before this shard's first
	// one, and `chunk_base` chunks are closed already; the ranks take their turns in stream order)
	int64_t token[2] = {0, 0};
	if(sc && (rc = shard_chunk_token(sc, true, token))) return rc;
	const int64_t chunk_base = token[1];
	std::vector<int64_t> starts{-token[0]};
	{
		int64_t filed = 0, in_chunk = token[0];
		for(int64_t k = 0; k < n_slots;) {
			if(k + 8 <= n_slots) {          // eight slots at once while no chunk can close among them (a slot counts 2 at most)
				uint64_t v;
				memcpy(&v, &cnt8[(size_t) k], 8);
				const int64_t sum = (int64_t) ((v * 0x0101010101010101ull) >> 56);
				if(in_chunk + sum < mf) { filed += sum; in_chunk += sum; k += 8; continue; }
			}
			const int c = cnt8[(size_t) k++];
			if(!c) continue;
			filed += c; in_chunk += c;
			if(in_chunk >= mf) { starts.push_back(filed); in_chunk = 0; }
		}
		token[0] = in_chunk; token[1] = chunk_base + (int64_t) starts.size() - 1;
	}
	if(sc && (rc = shard_chunk_token(sc, false, token))) return rc;
	stamp("fragment count + chunks");
	const int64_t *d_starts = nullptr;
	if((rc = B.up(starts.data(), starts.size(), 1, &d_starts))) return rc;
	int64_t *f_src, *f_rank, *f_name = nullptr;
	int32_t *f_rc, *f_t, *f_nh;
	if((rc = B.get((size_t) nf + 1, &f_src)) || (rc = B.get((size_t) nf + 1, &f_rank)) || (rc = B.get((size_t) nf + 1, &f_rc)) || (rc = B.get((size_t) nf + 1, &f_t)) ||
	   (rc = B.get((size_t) nf + 1, &f_nh))) return rc;
	kmahip_reads dF{};
	kmahip_traces tr;
	memset(&tr, 0, sizeof tr);
	if(nf > 0) {
		hipLaunchKernelGGL(pe_frag_fill_kernel, dim3((unsigned) ((n_slots + 255) / 256)), dim3(256), 0, s, n_slots, cc.tmpl, A.r_n, A.fr_read, A.fr_rc, f_off, d_starts,
		                   (int) starts.size(), mf, chunk_base, f_src, f_rc, f_t, f_nh, f_rank);
		HIP_TRY(hipGetLastError());
		if((rc = cp ? gather_batch2(B, dR, n, CR.d, f_src, nf, &dF, s) : gather_batch(B, dR, f_src, nf, &dF, s))) return rc;
		if(cp) {          // (the headers: a record's is its read's)
			if((rc = B.get((size_t) nf + 1, &f_name))) return rc;
			hipLaunchKernelGGL(pe_name_src_kernel, dim3((unsigned) ((nf + 255) / 256)), dim3(256), 0, s, nf, f_src, n, d_rec_read, f_name);
			HIP_TRY(hipGetLastError());
		}
		stamp("fragment batch");
		// stage 3c per fragment; the run pool is sized for a handful of runs per read and grown on demand
		if((rc = B.get((size_t) 10 * nf + 10, &tr.stats)) || (rc = B.get((size_t) nf + 1, &tr.ops_off)) || (rc = B.get((size_t) nf + 1, &tr.n_ops))) return rc;
		tr.ops_cap = 6 * nf + (1 << 20);
		for(int attempt = 0;; ++attempt) {
			if((rc = B.get((size_t) tr.ops_cap, &tr.ops))) return rc;
			if((rc = kmahip_launch_trace(db, ws, &dF, f_rc, f_t, d_ok, p, &tr, s))) return rc;
			HIP_TRY(hipStreamSynchronize(s));
			unsigned long long used = 0;
			const int st = ws_status(ws, &used);
			if(st == 2 || (int64_t) used > tr.ops_cap) {
				if(attempt >= 2) { kmahip_set_error("alignment run pool: %llu runs needed", used); return KMAHIP_EOVERFLOW; }
				tr.ops_cap = (int64_t) used + (1 << 20);
				continue;
			}
			if(st == 16 && grow_mem_cap(ws)) { --attempt; continue; }
			if(st) { kmahip_set_error("trace stage: a read needs more scratch than the workspace holds (status %d)", st); return KMAHIP_EDEVICE; }
			break;
		}
	}
	out->ms[3] = since(t);
	if(sc) {
		// exchange 3 and the owners' work: the filed fragments travel with their positions (shard_finish)
		std::vector<int64_t> src((size_t) nf + 1);
		if(nf) HIP_TRY(hipMemcpy(src.data(), f_name ? f_name : f_src, (size_t) nf * 8, hipMemcpyDeviceToHost));
		if(nf == 0) {          // (an empty fragment batch still takes part in the exchanges)
			if((rc = B.get(1, const_cast<uint64_t **>(&dF.seq), true)) || (rc = B.get(2, const_cast<int64_t **>(&dF.seq_off), true)) || (rc = B.get(1, const_cast<int32_t **>(&dF.len), true)) ||
			   (rc = B.get(1, const_cast<int32_t **>(&dF.N), true)) || (rc = B.get(2, const_cast<int64_t **>(&dF.N_off), true)) ||
			   (rc = B.get(16, &tr.stats, true)) || (rc = B.get(2, &tr.ops_off, true)) || (rc = B.get(2, &tr.n_ops, true)) || (rc = B.get(2, &tr.ops, true))) return rc;
			dF.n_reads = 0; dF.max_len = R.max_len;
		}
		return shard_pe_tail(sc, db, ws, B, batch, dF, f_rc, f_t, f_nh, f_rank, tr, src.data(), out->rows, out->n_rows, mf + 1, t);
	}

	// stage 3c per template
	if(nf > 0) {
		kmahip_assemble_opts ao = {mf + 1, evalue, bcd, 0, out->caller, out->sig90, f_rank, out->support};
		if((rc = kmahip_assemble2_dev(db, ws, &dF, f_rc, f_t, &tr, &ao, &out->assembly))) return rc;
	} else for(size_t i = 0; i < D; ++i) { out->assembly.cover[i] = 0; out->assembly.aln_len[i] = 0; out->assembly.depth[i] = 0; out->assembly.asm_len[i] = 0; }
	out->ms[4] = since(t);

	// `.frag`: the per-fragment columns come back; the reads and their headers are the host batch's, through the fragments' read numbers
	if(frag_path && nf > 0 && (pd || cp || !getenv("KMAHIP_PE_HOST_FRAG"))) {
		// the fragments and their figures are in HBM: the headers go up (a resident batch has them there), the rows are ordered and
		// formatted there (session.hip); the host compresses. (KMAHIP_PE_HOST_FRAG: the columns back and the rows made on the host, as in round 2)
		const char *d_names = pd ? pd->d_names : nullptr;
		const int64_t *d_name_off = pd ? pd->d_name_off : nullptr;
		if(!pd && ((rc = B.up(batch->names, (size_t) batch->name_off[n], 1, &d_names)) || (rc = B.up(batch->name_off, (size_t) n + 1, 0, &d_name_off)))) return rc;
		int64_t rows = 0;
		if(pinned.th.joinable()) pinned.th.join();
		if((rc = kmahip_frag_write_dev(db, &dF, d_names, d_name_off, f_name ? f_name : f_src, f_rc, f_t, f_nh, tr.stats, f_rank, mf + 1, frag_path, pd ? pd->text_chunk : pin_chunk,
		                               pd ? pd->h_text : (pinned.ok ? pinned.buf : nullptr), &rows))) return rc;
		if(pd && pd->frag_rows) *pd->frag_rows = rows;
	} else if(frag_path && nf > 0) {
		int32_t *stats4 = nullptr;
		if((rc = B.get((size_t) nf * 4 + 4, &stats4))) return rc;
		hipLaunchKernelGGL(pe_stats4_kernel, dim3((unsigned) ((nf + 255) / 256)), dim3(256), 0, s, nf, tr.stats, stats4);
		HIP_TRY(hipGetLastError());
		if(nf > n) { kmahip_set_error("more fragments than reads"); return KMAHIP_EDEVICE; }
		HC.wait();
		HIP_TRY(hipMemcpy(h_src, f_src, (size_t) nf * 8, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(h_rank, f_rank, (size_t) nf * 8, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(h_rc, f_rc, (size_t) nf * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(h_t, f_t, (size_t) nf * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(h_nh, f_nh, (size_t) nf * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(h_stats, stats4, (size_t) nf * 16, hipMemcpyDeviceToHost));
		stamp("fragment columns back");
		int64_t rows = 0;
		if((rc = kmahip_frag_write_src(frag_path, db, &R, nf, h_src, h_rc, h_t, h_nh, h_stats, 4, mf + 1, 0, h_rank,
		                               batch->names, batch->name_off, &rows))) return rc;
	} else if(frag_path) {          // nothing filed: the file is written all the same (an empty gzip stream)
		const int64_t none64 = 0;
		const int32_t none[4] = {0, 0, 0, 0};
		int64_t rows = 0;
		if((rc = kmahip_frag_write_src(frag_path, db, &R, 0, &none64, none, none, none, none, 4, mf + 1, 0, &none64, "", &none64, &rows))) return rc;
	}
	out->ms[5] = since(t);
	return KMAHIP_OK;
}


// ---- the default mode (no -1t1): the chain finder's records, put in stream order and made a batch of their own on the device, then
// every record through the stages of kmahip_run_se with its query bounds ------------------------------------------------------------
int kmahip_chain_device(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *d, const kmahip_params *p, const kmahip_chain_params *cp, int32_t *rec, int64_t *rec_T,
                        int32_t *T, int64_t rec_cap, int64_t T_cap, int64_t *n_recs, int64_t *n_T);                            // chain.hip

namespace {

__device__ __forceinline__ int64_t crec_read(const int32_t *r) { return (int64_t) (uint32_t) r[0] | ((int64_t) r[1] << 32); }

__global__ __launch_bounds__(256) void chain_count_kernel(int64_t m, const int32_t *rec, unsigned long long *cnt) {
	const int64_t x = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(x < m) atomicAdd(&cnt[crec_read(rec + 8 * x)], 1ull);
}

// stream order: reads ascending, a read's chains by their ordinal (0, 1, ... per read)
__global__ __launch_bounds__(256) void chain_place_kernel(int64_t m, const int32_t *rec, const int64_t *first, int64_t *o_read, int32_t *o_rcflag, int32_t *o_emit,
                                                          int32_t *o_qs, int32_t *o_qe, int64_t *o_nT, int64_t *slot_of) {
	const int64_t x = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(x > m) return;
	if(x == m) { o_nT[m] = 0; return; }
	const int32_t *r = rec + 8 * x;
	const int64_t rd = crec_read(r), at = first[rd] + r[2];
	o_read[at] = rd; o_rcflag[at] = r[3]; o_emit[at] = r[4]; o_qs[at] = r[5]; o_qe[at] = r[6]; o_nT[at] = r[7]; slot_of[at] = x;
}

__global__ __launch_bounds__(256) void chain_lists_kernel(int64_t m, const int64_t *slot_of, const int64_t *rec_T, const int32_t *T, const int64_t *T_off, int32_t *o_T) {
	const int64_t x = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(x >= m) return;
	const int32_t *src = T + rec_T[slot_of[x]];
	int32_t *dst = o_T + T_off[x];
	const int nT = (int) (T_off[x + 1] - T_off[x]);
	for(int i = 0; i < nT; ++i) dst[i] = src[i];
}

// gather_copy_kernel where a record may print the reverse complement of its read (rc_comp, compdna.c:228-256: the bits complemented,
// an N keeps its place from the other end)
__global__ __launch_bounds__(256) void gather_copy_rc_kernel(int64_t m, const int64_t *idx, const int32_t *turn, const uint64_t *seq, const int64_t *seq_off,
                                                             const int32_t *len, const int32_t *N, const int64_t *N_off, uint64_t *o_seq, const int64_t *o_seq_off,
                                                             int32_t *o_len, int32_t *o_N, const int64_t *o_N_off) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= m) return;
	const int64_t r = idx[i];
	const int L = len[r], w = (L + 31) >> 5;
	const uint64_t *a = seq + seq_off[r];
	uint64_t *b = o_seq + o_seq_off[i];
	const int32_t *na = N + N_off[r];
	int32_t *nb = o_N + o_N_off[i];
	const int nn = (int) (N_off[r + 1] - N_off[r]);
	o_len[i] = L;
	b[w] = 0;
	if(!turn[i]) {
		for(int x = 0; x < w; ++x) b[x] = a[x];
		for(int x = 0; x < nn; ++x) nb[x] = na[x];
		return;
	}
	// word x of the turned read = bases L-1-32x downwards; taken two source words at a time: the 64 bits that end at base q0 = L-1-32x,
	// complemented, their 2-bit groups reversed
	for(int x = 0; x < w; ++x) {
		const int q0 = L - 1 - 32 * x;                 // last source base of this word (>= 0)
		const int wi = q0 >> 5, sh = 2 * (31 - (q0 & 31));   // bits of word wi behind base q0
		uint64_t v = a[wi] >> sh;                      // base q0 in the lowest two bits
		if(sh && wi > 0) v |= a[wi - 1] << (64 - sh);
		v = ~v;
		// reverse the 2-bit groups
		v = ((v >> 2) & 0x3333333333333333ull) | ((v & 0x3333333333333333ull) << 2);
		v = ((v >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((v & 0x0F0F0F0F0F0F0F0Full) << 4);
		v = __builtin_bswap64(v);
		const int have = q0 + 1 < 32 ? q0 + 1 : 32;     // bases in this word
		if(have < 32) v &= ~0ull << (2 * (32 - have));
		b[x] = v;
	}
	for(int x = 0; x < nn; ++x) nb[x] = L - 1 - na[nn - 1 - x];
}

}  // namespace

static int chain_records(kmahip_db *db, kmahip_ws *ws, DevBlock &B, const kmahip_reads &dR, const kmahip_reads *reads, const kmahip_params *p,
                         const kmahip_chain_params *cp, ChainRecs &R, const std::function<void(const char *)> &lap) {
	const int64_t n = dR.n_reads;
	hipStream_t s = 0;
	int rc;
	// stage 2: one record per accepted chain, in no particular order
	int32_t *rec = nullptr, *T = nullptr;
	int64_t *rec_T = nullptr;
	int64_t rec_cap = 2 * n + 1024, T_cap = 16 * n + 4096, m = 0, n_T = 0;
	for(int attempt = 0; n > 0; ++attempt) {
		if((rc = B.get((size_t) rec_cap * 8, &rec)) || (rc = B.get((size_t) rec_cap, &rec_T)) || (rc = B.get((size_t) T_cap, &T))) return rc;
		rc = kmahip_chain_device(db, ws, &dR, p, cp, rec, rec_T, T, rec_cap, T_cap, &m, &n_T);
		if(rc == KMAHIP_EOVERFLOW && attempt < 3 && (m > rec_cap || n_T > T_cap)) {
			rec_cap = std::max(rec_cap, m + 16); T_cap = std::max(T_cap, n_T + 16);
			continue;
		}
		if(rc) return rc;
		break;
	}
	lap("stage 2 (chain_kernel)");
	// the records in stream order
	unsigned long long *cnt;
	int64_t *first, *o_read, *o_nT, *slot_of, *T_off;
	int32_t *o_rcflag, *o_emit, *o_qs, *o_qe, *o_T, *zero;
	if((rc = B.get((size_t) n + 1, &cnt, true)) || (rc = B.get((size_t) n + 1, &first)) || (rc = B.get((size_t) m + 1, &o_read)) || (rc = B.get((size_t) m + 1, &o_nT)) ||
	   (rc = B.get((size_t) m + 1, &slot_of)) || (rc = B.get((size_t) m + 1, &T_off, true)) || (rc = B.get((size_t) m + 1, &o_rcflag)) || (rc = B.get((size_t) m + 1, &o_emit)) ||
	   (rc = B.get((size_t) m + 1, &o_qs)) || (rc = B.get((size_t) m + 1, &o_qe)) || (rc = B.get((size_t) n_T + 1, &o_T)) || (rc = B.get((size_t) m + 1, &zero, true))) return rc;
	kmahip_reads d{};
	if(m) {
		hipLaunchKernelGGL(chain_count_kernel, dim3((unsigned) ((m + 255) / 256)), dim3(256), 0, s, m, rec, cnt);
		if((rc = scan_i64(B, (const int64_t *) cnt, first, (size_t) n + 1, s))) return rc;
		hipLaunchKernelGGL(chain_place_kernel, dim3((unsigned) ((m + 256) / 256)), dim3(256), 0, s, m, rec, first, o_read, o_rcflag, o_emit, o_qs, o_qe, o_nT, slot_of);
		if((rc = scan_i64(B, o_nT, T_off, (size_t) m + 1, s))) return rc;
		hipLaunchKernelGGL(chain_lists_kernel, dim3((unsigned) ((m + 255) / 256)), dim3(256), 0, s, m, slot_of, rec_T, T, T_off, o_T);
		HIP_TRY(hipGetLastError());
		// ... as a batch of their own: the read, or its reverse complement where the record prints that, with its bounds
		int64_t *wc, *nc, *so, *no;
		if((rc = B.get((size_t) m + 1, &wc)) || (rc = B.get((size_t) m + 1, &nc)) || (rc = B.get((size_t) m + 1, &so)) || (rc = B.get((size_t) m + 1, &no))) return rc;
		hipLaunchKernelGGL(gather_sizes_kernel, dim3((unsigned) ((m + 256) / 256)), dim3(256), 0, s, m, o_read, dR.len, dR.N_off, wc, nc);
		if((rc = scan_i64(B, wc, so, (size_t) m + 1, s)) || (rc = scan_i64(B, nc, no, (size_t) m + 1, s))) return rc;
		int64_t tw = 0, tn = 0;
		HIP_TRY(hipMemcpyAsync(&tw, so + m, 8, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(&tn, no + m, 8, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		uint64_t *seq;
		int32_t *len, *N;
		if((rc = B.get((size_t) tw + 2, &seq)) || (rc = B.get((size_t) m + 1, &len)) || (rc = B.get((size_t) tn + 1, &N))) return rc;
		HIP_TRY(hipMemsetAsync(seq + tw, 0, 16, s));
		HIP_TRY(hipMemsetAsync(len + m, 0, 4, s));
		hipLaunchKernelGGL(gather_copy_rc_kernel, dim3((unsigned) ((m + 255) / 256)), dim3(256), 0, s, m, o_read, o_emit, dR.seq, dR.seq_off, dR.len, dR.N, dR.N_off,
		                   seq, so, len, N, no);
		HIP_TRY(hipGetLastError());
		d.n_reads = m; d.seq = seq; d.seq_off = so; d.len = len; d.N = N; d.N_off = no; d.seq_words = tw; d.N_total = tn; d.max_len = reads->max_len;
		d.q_start = o_qs; d.q_end = o_qe;
		HIP_TRY(hipStreamSynchronize(s));
	}
	lap("records in stream order, record batch");
	R.m = m; R.n_T = n_T; R.d = d; R.o_read = o_read; R.o_emit = o_emit;
	R.c.rc_flag = o_rcflag; R.c.flag = zero; R.c.T_off = T_off; R.c.T = o_T; R.c.T_cap = n_T + 1;
	return KMAHIP_OK;
}

int kmahip_chain_records_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *dR, const kmahip_params *p, const kmahip_chain_params *cp, KmaChainRecs *out) {
	DevBlock *B = new DevBlock();
	B->expect((size_t) dR->seq_words * 8 + (size_t) dR->N_total * 4 + (size_t) dR->n_reads * 200 + (16u << 20));
	ChainRecs CR;
	const int rc = chain_records(db, ws, *B, *dR, dR, p, cp, CR, [](const char *) {});
	if(rc) { delete B; return rc; }
	out->m = CR.m; out->n_T = CR.n_T; out->d = CR.d; out->c = CR.c; out->o_read = CR.o_read; out->o_emit = CR.o_emit; out->block = B;
	return KMAHIP_OK;
}
void kmahip_chain_records_free(KmaChainRecs *r) {
	if(r && r->block) { delete (DevBlock *) r->block; r->block = nullptr; }
}

extern "C" int kmahip_run_chain(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const char *names, const int64_t *name_off,
                                const kmahip_params *p, const kmahip_chain_params *cp, double evalue, int bcd, int64_t max_frag,
                                const char *frag_path, kmahip_run *out) {
	if(!db || !ws || !reads || !p || !out || !out->rows || !out->assembly.cover || !out->assembly.aln_len || !out->assembly.depth || !out->assembly.asm_len) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	if(frag_path && reads && reads->n_reads > 0 && (!names || !name_off)) { kmahip_set_error("the fragment file needs the read headers"); return KMAHIP_EINVAL; }
	const int64_t n = reads->n_reads;
	if(n < 0 || reads->seq_words < 0 || reads->N_total < 0) { kmahip_set_error("negative size"); return KMAHIP_EINVAL; }
	const size_t D = db->info.DB_size;
	for(int i = 0; i < 6; ++i) out->ms[i] = 0;
	out->n_rows = 0;
	hipStream_t s = 0;
	auto t = std::chrono::steady_clock::now();
	const bool dbg = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
	auto t_lap = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) { if(dbg) fprintf(stderr, "[kmahip] run_chain: %s %.1f ms\n", what, since(t_lap)); };
	int rc;
	// the batch, once
	DevBlock B;
	B.expect((size_t) reads->seq_words * 16 + (size_t) reads->N_total * 8 + (size_t) n * 480 + (64u << 20));
	kmahip_reads dR = *reads;
	dR.q_start = nullptr; dR.q_end = nullptr;
	if((rc = B.up(reads->seq, (size_t) reads->seq_words, 2, &dR.seq)) || (rc = B.up(reads->seq_off, (size_t) n + 1, 0, &dR.seq_off)) ||
	   (rc = B.up(reads->len, (size_t) n, 1, &dR.len)) || (rc = B.up(reads->N, (size_t) reads->N_total, 1, &dR.N)) ||
	   (rc = B.up(reads->N_off, (size_t) n + 1, 0, &dR.N_off))) return rc;
	HIP_TRY(hipStreamSynchronize(s));
	out->ms[0] = since(t);
	lap("reads uploaded");
	ChainRecs CR;
	if((rc = chain_records(db, ws, B, dR, reads, p, cp, CR, lap))) return rc;
	const int64_t m = CR.m, n_T = CR.n_T;
	kmahip_reads d = CR.d;
	int64_t *o_read = CR.o_read;
	int32_t *o_emit = CR.o_emit;
	kmahip_cands c = CR.c;
	HostCols H;
	int32_t *k_tmpl = H.get<int32_t>((size_t) m + 1), *k_nh = H.get<int32_t>((size_t) m + 1), *k_rc = H.get<int32_t>((size_t) m + 1), *k_stats = H.get<int32_t>((size_t) m * 10 + 10);
	int64_t *h_read = H.get<int64_t>((size_t) m + 1);
	int32_t *h_emit = H.get<int32_t>((size_t) m + 1);
	if(!k_tmpl || !k_nh || !k_rc || !k_stats || !h_read || !h_emit) { kmahip_set_error("out of host memory"); return KMAHIP_ENOMEM; }
	if(m > 100000) H.start();
	PerRead pr = {k_tmpl, k_nh, k_rc, k_stats};
	if(m) {
		if((rc = run_after_stage2(db, ws, B, d, c, n_T, p, evalue, bcd, max_frag, out, pr, t))) return rc;
	} else {
		std::vector<uint64_t> w(D, 0);
		if((rc = kmahip_res_rows(db, w.data(), evalue, p->scoreT, out->rows, out->rows_cap, &out->n_rows))) return rc;
		for(size_t i = 0; i < D; ++i) { out->assembly.cover[i] = 0; out->assembly.aln_len[i] = 0; out->assembly.depth[i] = 0; out->assembly.asm_len[i] = 0; }
	}
	lap("stages 3a ... pile-up");
	if(frag_path && m) {
		// the rows are formatted from the reads as they came: a record that printed the reverse complement turns its row once more
		HIP_TRY(hipMemcpy(h_read, o_read, (size_t) m * 8, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(h_emit, o_emit, (size_t) m * 4, hipMemcpyDeviceToHost));
		for(int64_t x = 0; x < m; ++x) k_rc[(size_t) x] ^= h_emit[(size_t) x] & 1;
		int64_t rows = 0;
		if((rc = kmahip_frag_write_src(frag_path, db, reads, m, h_read, k_rc, k_tmpl, k_nh, k_stats, 10, max_frag, 0, nullptr,
		                               names, name_off, &rows))) return rc;
	}
	else if(frag_path) {          // no record: the file is written all the same (an empty gzip stream)
		const int64_t none64 = 0;
		const int32_t none[10] = {0};
		int64_t rows = 0;
		if((rc = kmahip_frag_write_src(frag_path, db, reads, 0, &none64, none, none, none, none, 10, max_frag, 0, nullptr, names ? names : "", name_off ? name_off : &none64, &rows))) return rc;
	}
	lap("fragment file");
	out->ms[5] += since(t);
	return KMAHIP_OK;
}

// ---- `-Mt1 t`: raw reads straight to stage 3c against one template (runKMA_Mt1, mt1.c:86-500) ------------------------------
extern "C" int kmahip_run_mt1(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, int32_t tmpl, int one2one, const kmahip_params *p,
                              const kmahip_assemble_opts *aopts, kmahip_run *out) {
	if(!db || !ws || !reads || !p || !aopts || !out || !out->rows || out->rows_cap < 1 || !out->assembly.cover || !out->assembly.aln_len || !out->assembly.depth || !out->assembly.asm_len) {
		kmahip_set_error("null argument"); return KMAHIP_EINVAL;
	}
	const int64_t n = reads->n_reads;
	if(n < 0 || reads->seq_words < 0 || reads->N_total < 0) { kmahip_set_error("negative size"); return KMAHIP_EINVAL; }
	const size_t D = db->info.DB_size;
	if(tmpl < 1 || (size_t) tmpl >= D) { kmahip_set_error("template %d out of range", tmpl); return KMAHIP_EINVAL; }
	for(int i = 0; i < 6; ++i) out->ms[i] = 0;
	out->n_rows = 0;
	hipStream_t s = 0;
	DevBlock B;
	int rc;
	auto t = std::chrono::steady_clock::now();
	kmahip_reads d = *reads;
	d.q_start = nullptr; d.q_end = nullptr;       // (host pointers, if any: this entry point maps whole reads)
	if((rc = B.up(reads->seq, (size_t) reads->seq_words, 2, &d.seq)) || (rc = B.up(reads->seq_off, (size_t) n + 1, 0, &d.seq_off)) ||
	   (rc = B.up(reads->len, (size_t) n, 1, &d.len)) || (rc = B.up(reads->N, (size_t) reads->N_total, 1, &d.N)) ||
	   (rc = B.up(reads->N_off, (size_t) n + 1, 0, &d.N_off))) return rc;
	HIP_TRY(hipStreamSynchronize(s));
	out->ms[0] = since(t);
	// stage 3c per read: strand + traceback; the run pool grows on demand
	kmahip_traces tr;
	int32_t *d_rc = nullptr, *d_tmpl = nullptr;
	if((rc = B.get((size_t) 10 * n + 10, &tr.stats)) || (rc = B.get((size_t) n + 1, &tr.ops_off)) || (rc = B.get((size_t) n + 1, &tr.n_ops)) ||
	   (rc = B.get((size_t) n + 1, &d_rc)) || (rc = B.get((size_t) n + 1, &d_tmpl))) return rc;
	int64_t total_bases = 0;
	for(int64_t i = 0; i < n; ++i) total_bases += reads->len[i];
	tr.ops_cap = total_bases / 3 + 8 * n + (1 << 16);
	for(int attempt = 0; n; ++attempt) {
		if((rc = B.get((size_t) tr.ops_cap, &tr.ops))) return rc;
		if((rc = kmahip_launch_longtrace(db, ws, &d, nullptr, tmpl, nullptr, nullptr, one2one, p, &tr, d_rc, s))) return rc;
		unsigned long long used = 0;
		const int st = ws_status(ws, &used);
		if(st == 2 || (int64_t) used > tr.ops_cap) {
			if(attempt >= 2) { kmahip_set_error("alignment run pool: %llu runs needed", used); return KMAHIP_EOVERFLOW; }
			tr.ops_cap = (int64_t) used + (1 << 16);
			continue;
		}
		break;
	}
	out->ms[3] = since(t);
	// the per-read figures: Score of the `.res` row = sum of KMA()'s own scores of the kept reads (alnToMat, assembly.c:1328-1334),
	// i.e. without the end bonus the read filter added
	std::vector<int32_t> stats((size_t) 10 * n + 10, 0), h_tmpl((size_t) n + 1, 0);
	if(n) HIP_TRY(hipMemcpy(stats.data(), tr.stats, (size_t) n * 40, hipMemcpyDeviceToHost));
	const int t_len = db->h_tlen[(size_t) tmpl];
	uint64_t score = 0;
	for(int64_t i = 0; i < n; ++i) {
		const int32_t *st = &stats[(size_t) 10 * i];
		if(st[3] == 0) continue;
		h_tmpl[(size_t) i] = tmpl;
		score += (uint64_t) (st[0] - p->rw.Wl * ((st[1] == 0) + (st[2] == t_len)));
	}
	if(n) HIP_TRY(hipMemcpy(d_tmpl, h_tmpl.data(), (size_t) n * 4, hipMemcpyHostToDevice));
	kmahip_res_row &row = out->rows[0];
	row.template_id = tmpl; row.template_length = t_len; row.score = score; row.expected = 0;
	row.q_value = (double) score; row.p_value = kmahip_p_chisqr((long double) score);
	row.significant = kmahip_cmp(row.p_value <= aopts->evalue && score > 0, (double) score >= p->scoreT * t_len);     // mt1.c:419
	out->n_rows = 1;
	out->ms[2] = since(t);
	kmahip_assemble_opts ao = *aopts;
	ao.order = 1;
	if(n && score) {
		if((rc = kmahip_assemble2_dev(db, ws, &d, d_rc, d_tmpl, &tr, &ao, &out->assembly))) return rc;
	} else for(size_t i = 0; i < D; ++i) { out->assembly.cover[i] = 0; out->assembly.aln_len[i] = 0; out->assembly.depth[i] = 0; out->assembly.asm_len[i] = 0; }
	out->ms[4] = since(t);
	if(n) {
		if(out->tmpl) memcpy(out->tmpl, h_tmpl.data(), (size_t) n * 4);
		if(out->n_hits) for(int64_t i = 0; i < n; ++i) out->n_hits[i] = 1;
		if(out->rc) HIP_TRY(hipMemcpy(out->rc, d_rc, (size_t) n * 4, hipMemcpyDeviceToHost));
		if(out->trace_stats) memcpy(out->trace_stats, stats.data(), (size_t) n * 40);
	}
	out->ms[5] = since(t);
	return KMAHIP_OK;
}


// ---- the single-end run over read shards, one process per GPU (kmahip.h: kmahip_run_se_sharded) -----------------------
namespace {

constexpr int ROW = 20;        // int32 fields of a travelling read: frag_rank lo / hi, len, rc, tmpl, n_hits, nN, n_ops, stats[10], 2 spare

// ConClave's per-template outputs side by side as u64, for ONE all-reduce: w_scores | depth | fragment counts | read counts
__global__ __launch_bounds__(256) void shard_pack_kernel(int64_t D, const uint64_t *w, const uint64_t *depth, const uint32_t *frags, const uint32_t *reads, uint64_t *out) {
	const int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(t >= D) return;
	out[t] = w[t]; out[D + t] = depth[t]; out[2 * D + t] = frags[t]; out[3 * D + t] = reads[t];
}

// filed flag per read (ConClave gave it a template) for the scan that numbers the filed fragments; destination of a kept read
__global__ __launch_bounds__(256) void shard_filed_kernel(int64_t n, const int32_t *tmpl, int64_t *filed) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i <= n) filed[i] = i < n && tmpl[i] != 0;
}
__global__ __launch_bounds__(256) void shard_add_kernel(int64_t n, int64_t *v, int64_t base) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i < n) v[i] += base;
}
__global__ __launch_bounds__(256) void shard_dest_kernel(int64_t n, const int32_t *tmpl, const int32_t *stats, const int32_t *owner, int world, uint32_t *dest, int64_t *idx) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n) return;
	const int t = tmpl[i];
	const int dd = (t != 0 && stats[10 * i + 3] != 0) ? owner[abs(t)] : world;
	dest[i] = (uint32_t) dd; idx[i] = i;
}

// where each destination's stretch begins in the sorted order (written only at the boundaries: no contended counters)
__global__ __launch_bounds__(256) void shard_bounds_kernel(int64_t n, const uint32_t *sorted, unsigned long long *first) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n) return;
	if(i == 0 || sorted[i] != sorted[i - 1]) first[sorted[i]] = (unsigned long long) i;
}

__global__ __launch_bounds__(256) void shard_rows_kernel(int64_t m, const int64_t *idx, const int64_t *frag_rank, const int32_t *len, const int32_t *rc,
                                                         const int32_t *tmpl, const int32_t *n_hits, const int64_t *N_off, const int32_t *n_ops, const int32_t *stats,
                                                         int32_t *rows, int64_t *ops_cnt) {
	const int64_t x = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(x > m) return;
	if(x == m) { ops_cnt[m] = 0; return; }
	const int64_t i = idx[x];
	int32_t *r = rows + ROW * x;
	const int64_t fr = frag_rank[i];
	r[0] = (int32_t) (fr & 0xFFFFFFFFll); r[1] = (int32_t) (fr >> 32);
	r[2] = len[i]; r[3] = rc[i]; r[4] = tmpl[i]; r[5] = n_hits[i]; r[6] = (int32_t) (N_off[i + 1] - N_off[i]); r[7] = n_ops[i];
	for(int k = 0; k < 10; ++k) r[8 + k] = stats[10 * i + k];
	r[18] = 0; r[19] = 0;
	ops_cnt[x] = n_ops[i];
}

__global__ __launch_bounds__(256) void shard_ops_kernel(int64_t m, const int64_t *idx, const int64_t *ops_off, const int32_t *n_ops, const uint32_t *ops, const int64_t *o_off, uint32_t *o_ops) {
	const int64_t x = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(x >= m) return;
	const int64_t i = idx[x];
	const uint32_t *src = ops + ops_off[i];
	uint32_t *dst = o_ops + o_off[x];
	for(int k = 0; k < n_ops[i]; ++k) dst[k] = src[k];
}

// what an owner received: the rows back into per-read columns, and the sizes the CSR offsets are scanned from
__global__ __launch_bounds__(256) void shard_unpack_kernel(int64_t m, const int32_t *rows, int64_t *frag_rank, int32_t *len, int32_t *rc, int32_t *tmpl, int32_t *n_hits,
                                                           int32_t *n_ops, int32_t *stats, int64_t *words, int64_t *n_N, int64_t *ops_cnt) {
	const int64_t x = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(x > m) return;
	if(x == m) { words[m] = 0; n_N[m] = 0; ops_cnt[m] = 0; return; }
	const int32_t *r = rows + ROW * x;
	frag_rank[x] = (int64_t) (uint32_t) r[0] | ((int64_t) r[1] << 32);
	len[x] = r[2]; rc[x] = r[3]; tmpl[x] = r[4]; n_hits[x] = r[5]; n_ops[x] = r[7];
	for(int k = 0; k < 10; ++k) stats[10 * x + k] = r[8 + k];
	words[x] = ((r[2] + 31) >> 5) + 1; n_N[x] = r[6]; ops_cnt[x] = r[7];
}

// files <prefix>.part0<ext> .. <prefix>.part<world - 1><ext> back to back into <prefix><ext> (behind `head`), the parts removed
int concat_parts(const std::string &prefix, const char *ext, int world, const char *head) {
	const std::string path = prefix + ext;
	FILE *o = fopen(path.c_str(), "wb");
	if(!o) { kmahip_set_error("cannot create %s", path.c_str()); return KMAHIP_EIO; }
	if(head) fputs(head, o);
	std::vector<char> buf(4u << 20);
	int rc = KMAHIP_OK;
	for(int r = 0; r < world && !rc; ++r) {
		const std::string part = prefix + ".part" + std::to_string(r) + ext;
		FILE *f = fopen(part.c_str(), "rb");
		if(!f) { kmahip_set_error("part %s is missing", part.c_str()); rc = KMAHIP_EIO; break; }
		size_t got;
		while((got = fread(buf.data(), 1, buf.size(), f)) > 0) if(fwrite(buf.data(), 1, got, o) != got) { kmahip_set_error("write to %s failed", path.c_str()); rc = KMAHIP_EIO; break; }
		fclose(f);
		remove(part.c_str());
	}
	if(fclose(o) != 0 && !rc) { kmahip_set_error("write to %s failed", path.c_str()); rc = KMAHIP_EIO; }
	return rc;
}

}  // namespace

// the 2-bit template store on the host (the t line of the `.aln` file): <prefix>.seq.b, read when first asked for
static int load_tseq(kmahip_db *db) {
	if(!db->h_tseq.empty()) return KMAHIP_OK;
	const size_t D = db->info.DB_size;
	if(db->h_tlen.size() != D) { kmahip_set_error("index has no .length.b / .seq.b"); return KMAHIP_EINVAL; }
	db->h_tseq_off.assign(D + 1, 0);
	for(size_t i = 2; i <= D; ++i) db->h_tseq_off[i] = db->h_tseq_off[i - 1] + (db->h_tlen[i - 1] >> 5) + 1;      // runkma.c:214-220
	std::vector<uint64_t> w((size_t) db->h_tseq_off[D] + 2, 0);
	FILE *f = fopen((db->prefix + ".seq.b").c_str(), "rb");
	if(!f || fread(w.data(), 8, (size_t) db->h_tseq_off[D], f) != (size_t) db->h_tseq_off[D]) { if(f) fclose(f); kmahip_set_error("cannot read %s.seq.b", db->prefix.c_str()); return KMAHIP_EIO; }
	fclose(f);
	db->h_tseq.swap(w);
	return KMAHIP_OK;
}

// One template's block of the `.aln` file as printConsensus writes it (printconsensus.c:26-37): "# name", then per 60 columns the
// lines "template:", the match line and "query:" (callConsensus assembly.c:1543-1611: t = the template's base, '-' at an insertion
// column; s = '|' where a call equals it, else '_'), of the alignment as assemble_KMA trims it (:2094-2119: the insertion columns
// called as gaps are gone). `cons`: the template's consensus string made with kmahip_assemble_opts.caller + 32 (every insertion
// column's character carries bit 7). Returns the length of the text written to `out`, -1 when `cap` is too small or on an error.
extern "C" int64_t kmahip_aln_entry(kmahip_db *db, int32_t tmpl, const char *name, const char *cons, char *out, int64_t cap) {
	if(!db || !name || !cons || !out || tmpl < 1 || (size_t) tmpl >= db->info.DB_size) { kmahip_set_error("kmahip_aln_entry: bad argument"); return -1; }
	if(load_tseq(db)) return -1;
	const uint64_t *ts = db->h_tseq.data() + db->h_tseq_off[(size_t) tmpl];
	const int t_len = db->h_tlen[(size_t) tmpl];
	std::string t, sl, q;
	int p = 0;
	for(const unsigned char *c = (const unsigned char *) cons; *c; ++c) {
		const bool ins = (*c & 0x80) != 0;
		const char b = (char) (*c & 0x7F);
		if(ins) {
			if(b == '-' || b == '_') continue;          // trimmed
			t.push_back('-'); q.push_back(b); sl.push_back('_');
			continue;
		}
		if(p >= t_len) { kmahip_set_error("kmahip_aln_entry: the consensus string has more template columns than template %d has bases (made without caller + 32 ?)", tmpl); return -1; }
		const char tb = "ACGT"[(ts[p >> 5] >> (62 - ((p & 31) << 1))) & 3ull];
		++p;
		t.push_back(tb); q.push_back(b);
		sl.push_back(b != '-' && tb == (char) toupper((unsigned char) b) ? '|' : '_');
	}
	std::string text = "# ";
	text += name; text += "\n";
	char head[3][16];
	snprintf(head[0], sizeof head[0], "%-10s\t", "template:"); snprintf(head[1], sizeof head[1], "%-10s\t", ""); snprintf(head[2], sizeof head[2], "%-10s\t", "query:");
	for(size_t i = 0; i < t.size(); i += 60) {
		const size_t n = std::min<size_t>(60, t.size() - i);
		text += head[0]; text.append(t, i, n); text += "\n";
		text += head[1]; text.append(sl, i, n); text += "\n";
		text += head[2]; text.append(q, i, n); text += "\n\n";
	}
	if((int64_t) text.size() > cap) { kmahip_set_error("kmahip_aln_entry: %zu bytes needed, room for %lld", text.size(), (long long) cap); return -1; }
	memcpy(out, text.data(), text.size());
	return (int64_t) text.size();
}

// The rows of `.res` (runkma.c:792-809), the entries of the consensus FASTA (printConsensus, printconsensus.c:38-60: the
// consensus line without its '-' columns, 60 per line) and, with aln_path, the blocks of the `.aln` file (:26-37; the consensus
// strings were then made with caller + 32) for the significant templates -- all of them, or those `owner` gives to
// `rank`. fsa_path NULL: no consensus file (-nc); aln_path NULL: no alignment file (-na).
int kmahip_write_res_fsa(kmahip_db *db, const char *res_path, const char *fsa_path, bool header, const kmahip_res_row *rows, int64_t n_rows,
                         const int32_t *owner, int rank, const int64_t *cover, const int64_t *aln_len, const int64_t *depth, const char *cons,
                         const int64_t *cons_off, double ID_t, double Depth_t, int ref_fsa, const char *aln_path) {
	int rc = kmahip_db_load_names(db);
	if(rc) return rc;
	FILE *res = fopen(res_path, "w"), *fsa = fsa_path ? fopen(fsa_path, "w") : nullptr, *aln = aln_path ? fopen(aln_path, "w") : nullptr;
	if(!res || (fsa_path && !fsa) || (aln_path && !aln)) { if(res) fclose(res); if(fsa) fclose(fsa); if(aln) fclose(aln); kmahip_set_error("cannot create %s", !res ? res_path : (fsa_path && !fsa) ? fsa_path : aln_path); return KMAHIP_EIO; }
	if(header) fputs("#Template\tScore\tExpected\tTemplate_length\tTemplate_Identity\tTemplate_Coverage\tQuery_Identity\tQuery_Coverage\tDepth\tq_value\tp_value\n", res);
	std::vector<char> line((1 << 16) + 512), block;
	std::string entry;
	for(int64_t r = 0; r < n_rows && !rc; ++r) {
		const kmahip_res_row &row = rows[r];
		const size_t tt = (size_t) row.template_id;
		if((owner && owner[tt] != rank) || !row.significant || tt - 1 >= db->h_names.size()) continue;
		const std::string &name = db->h_names[tt - 1];
		if(!kmahip_res_line(name.c_str(), &row, cover[tt], aln_len[tt], depth[tt], ID_t, Depth_t, line.data(), (int64_t) line.size())) continue;
		fputs(line.data(), res);
		const char *q0 = cons_off[tt] >= 0 ? cons + cons_off[tt] : "";
		if(aln) {
			const size_t len = strlen(q0);
			block.resize((len / 60 + 2) * 224 + name.size() + 16);
			const int64_t got = kmahip_aln_entry(db, (int32_t) tt, name.c_str(), q0, block.data(), (int64_t) block.size());
			if(got < 0) { rc = KMAHIP_EINVAL; break; }
			fwrite(block.data(), 1, (size_t) got, aln);
		}
		if(!fsa) continue;
		entry.clear();
		entry += ">"; entry += name; entry += "\n";
		int col = 0;
		// printConsensus (printconsensus.c:38-60): gap columns left out;
		// with -ref_fsa 0 (ref_fsa == 2) the gaps of template positions stay; insertion columns called as gaps ('_': the caller was asked to
		// mark them; or, with every insertion column flagged by bit 7 for the `.aln` writer, a flagged gap) were trimmed from the alignment
		// before (assembly.c:748-752). With -ref_fsa refCaller leaves no gap at a template position.
		for(const char *q = q0; *q; ++q) {
			const char b = (char) (*q & 0x7F);
			if(b == '_' || (b == '-' && (ref_fsa != 2 || (*q & 0x80)))) continue;
			entry.push_back(b);
			if(++col == 60) { entry.push_back('\n'); col = 0; }
		}
		if(col) entry.push_back('\n');
		fwrite(entry.data(), 1, entry.size(), fsa);
	}
	const bool bad = fclose(res) != 0;
	const bool bad2 = fsa && fclose(fsa) != 0, bad3 = aln && fclose(aln) != 0;
	if(rc) return rc;
	if(bad || bad2 || bad3) { kmahip_set_error("write to %s failed", res_path); return KMAHIP_EIO; }
	return KMAHIP_OK;
}

// Exchange 3 and everything behind it, for single-end and paired runs alike. `d`: the rank's n items in HBM (reads, or the filed
// fragments of a paired run) in the order of its part of the stream; tmpl (0: not filed), rc, n_hits, frag_rank (position among the
// filed fragments of the WHOLE stream, chunk arithmetic of the caller included) and the traces are per item, DEVICE pointers;
// name_src: which read of `batch` an item is (NULL: item i is read i). frag_counts: a per-template vector that is the same on every
// rank (the summed fragment counts), which the owners are cut by. chunk: the chunk length the pile-up and the writer divide
// frag_rank by.
static int shard_finish(kmahip_db *db, kmahip_ws *ws, kmahip_comm *comm, DevBlock &B, const kmahip_read_batch *batch, const kmahip_reads &d, const int32_t *i_tmpl,
                        const int32_t *i_rc, const int32_t *i_nhits, const int64_t *i_frag_rank, const kmahip_traces &tr, const int64_t *name_src,
                        const uint64_t *frag_counts, const kmahip_res_row *rows, int64_t n_rows, int64_t chunk, const kmahip_shard_opts *opts,
                        const char *out_prefix, double ms[8], std::chrono::steady_clock::time_point &t, int order = 0) {
	const int W = kmahip_comm_world(comm), rank = kmahip_comm_rank(comm);
	const int64_t n = d.n_reads;
	const size_t D = db->info.DB_size;
	hipStream_t s = 0;
	int rc;
	// exchange 3: every kept read to the owner of its template. Owners: contiguous template ranges, cut where the filed fragments
	// before a template reach the next 1 / W of all of them (the same on every rank: the counts are the summed ones).
	std::vector<int32_t> owner(D, 0);
	{
		unsigned long long tot = 0, before = 0;
		for(size_t tt = 0; tt < D; ++tt) tot += frag_counts[tt];
		for(size_t tt = 0; tt < D; ++tt) {
			owner[tt] = tot ? (int32_t) std::min<unsigned long long>((unsigned long long) (W - 1), (unsigned long long) ((unsigned __int128) before * (unsigned) W / tot)) : 0;
			before += frag_counts[tt];
		}
	}
	const int32_t *d_owner = nullptr;
	if((rc = B.up(owner.data(), D, 0, &d_owner))) return rc;
	int64_t *idx = nullptr, *idx2 = nullptr;
	uint32_t *dest = nullptr, *dest2 = nullptr;
	unsigned long long *d_cnt = nullptr;
	if((rc = B.get((size_t) n + 1, &idx)) || (rc = B.get((size_t) n + 1, &idx2)) || (rc = B.get((size_t) n + 1, &dest)) || (rc = B.get((size_t) n + 1, &dest2)) ||
	   (rc = B.get((size_t) W + 2, &d_cnt))) return rc;
	if(n) hipLaunchKernelGGL(shard_dest_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, n, i_tmpl, tr.stats, d_owner, W, dest, idx);
	HIP_TRY(hipGetLastError());
	// (the longest read of the run sizes the owners' scratch)
	std::vector<int64_t> all_meta((size_t) W);
	{
		const int64_t mine = (int64_t) d.max_len;
		if((rc = kmahip_comm_allgather(comm, &mine, sizeof mine, all_meta.data()))) return rc;
	}
	int max_len = 0;
	for(int r = 0; r < W; ++r) max_len = std::max(max_len, (int) all_meta[(size_t) r]);
	// kept reads ordered by destination, stream order inside one (a stable sort on the few bits of the destination)
	if(n) {
		size_t tmp_bytes = 0;
		int bits = 1;
		while((1 << bits) <= W) ++bits;
		if(rocprim::radix_sort_pairs(nullptr, tmp_bytes, dest, dest2, idx, idx2, (size_t) n, 0, (unsigned) bits, s) != hipSuccess) { kmahip_set_error("rocprim::radix_sort_pairs (size query) failed"); return KMAHIP_EDEVICE; }
		char *tmp = nullptr;
		if((rc = B.get(tmp_bytes, &tmp))) return rc;
		if(rocprim::radix_sort_pairs(tmp, tmp_bytes, dest, dest2, idx, idx2, (size_t) n, 0, (unsigned) bits, s) != hipSuccess) { kmahip_set_error("rocprim::radix_sort_pairs failed"); return KMAHIP_EDEVICE; }
	}
	std::vector<int64_t> seg((size_t) W + 1, 0);
	std::vector<unsigned long long> cnt((size_t) W + 1, 0);
	{
		std::vector<unsigned long long> first((size_t) W + 2, ~0ull);
		HIP_TRY(hipMemcpyAsync(d_cnt, first.data(), ((size_t) W + 2) * 8, hipMemcpyHostToDevice, s));
		if(n) hipLaunchKernelGGL(shard_bounds_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, n, dest2, d_cnt);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(first.data(), d_cnt, ((size_t) W + 2) * 8, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		first[(size_t) W + 1] = (unsigned long long) n;
		for(int r = W; r >= 0; --r) if(first[(size_t) r] == ~0ull) first[(size_t) r] = first[(size_t) r + 1];       // (a destination nobody goes to)
		for(int r = 0; r <= W; ++r) { if(r <= W) seg[(size_t) std::min(r, W)] = (int64_t) first[(size_t) r]; }
		for(int r = 0; r < W; ++r) cnt[(size_t) r] = first[(size_t) r + 1] - first[(size_t) r];
	}
	const int64_t m = seg[(size_t) W];
	kmahip_reads dK{};
	int32_t *rows_d = nullptr;
	int64_t *ops_cnt = nullptr, *ops_o = nullptr;
	uint32_t *ops_k = nullptr;
	if((rc = gather_batch(B, d, idx2, m, &dK, s))) return rc;
	if((rc = B.get((size_t) ROW * m + ROW, &rows_d)) || (rc = B.get((size_t) m + 1, &ops_cnt)) || (rc = B.get((size_t) m + 1, &ops_o))) return rc;
	hipLaunchKernelGGL(shard_rows_kernel, dim3((unsigned) ((m + 256) / 256)), dim3(256), 0, s, m, idx2, i_frag_rank, d.len, i_rc, i_tmpl, i_nhits, d.N_off, tr.n_ops,
	                   tr.stats, rows_d, ops_cnt);
	HIP_TRY(hipGetLastError());
	if((rc = scan_i64(B, ops_cnt, ops_o, (size_t) m + 1, s))) return rc;
	int64_t ops_total = 0;
	HIP_TRY(hipMemcpy(&ops_total, ops_o + m, 8, hipMemcpyDeviceToHost));
	if((rc = B.get((size_t) ops_total + 1, &ops_k))) return rc;
	if(m) hipLaunchKernelGGL(shard_ops_kernel, dim3((unsigned) ((m + 255) / 256)), dim3(256), 0, s, m, idx2, tr.ops_off, tr.n_ops, tr.ops, ops_o, ops_k);
	HIP_TRY(hipGetLastError());
	// block sizes per destination (rows, words, N positions, runs, name bytes), agreed on through the mailboxes
	std::vector<int64_t> h_idx((size_t) m + 1), so_at((size_t) W + 1), no_at((size_t) W + 1), oo_at((size_t) W + 1);
	if(m) HIP_TRY(hipMemcpy(h_idx.data(), idx2, (size_t) m * 8, hipMemcpyDeviceToHost));
	for(int r = 0; r <= W; ++r) {
		HIP_TRY(hipMemcpy(&so_at[(size_t) r], dK.seq_off + seg[(size_t) r], 8, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(&no_at[(size_t) r], dK.N_off + seg[(size_t) r], 8, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(&oo_at[(size_t) r], ops_o + seg[(size_t) r], 8, hipMemcpyDeviceToHost));
	}
	std::vector<char> name_send;
	std::vector<int64_t> name_at((size_t) W + 1, 0);
	for(int r = 0; r < W; ++r) {
		int64_t bytes = 0;
		for(int64_t x = seg[(size_t) r]; x < seg[(size_t) r + 1]; ++x) { const int64_t i = name_src ? name_src[h_idx[(size_t) x]] : h_idx[(size_t) x]; bytes += batch->name_off[i + 1] - batch->name_off[i]; }
		name_at[(size_t) r + 1] = name_at[(size_t) r] + bytes;
	}
	name_send.resize((size_t) name_at[(size_t) W] + 1);
	{
		char *o = name_send.data();
		for(int64_t x = 0; x < m; ++x) { const int64_t i = name_src ? name_src[h_idx[(size_t) x]] : h_idx[(size_t) x]; const int64_t l = batch->name_off[i + 1] - batch->name_off[i]; memcpy(o, batch->names + batch->name_off[i], (size_t) l); o += l; }
	}
	constexpr int NA = 5;          // arrays that travel
	std::vector<int64_t> mine((size_t) NA * W), all((size_t) NA * W * W);
	for(int r = 0; r < W; ++r) {
		mine[(size_t) (0 * W + r)] = (int64_t) cnt[(size_t) r] * ROW * 4;
		mine[(size_t) (1 * W + r)] = (so_at[(size_t) r + 1] - so_at[(size_t) r]) * 8;
		mine[(size_t) (2 * W + r)] = (no_at[(size_t) r + 1] - no_at[(size_t) r]) * 4;
		mine[(size_t) (3 * W + r)] = (oo_at[(size_t) r + 1] - oo_at[(size_t) r]) * 4;
		mine[(size_t) (4 * W + r)] = name_at[(size_t) r + 1] - name_at[(size_t) r];
	}
	if((rc = kmahip_comm_allgather(comm, mine.data(), mine.size() * 8, all.data()))) return rc;
	std::vector<int64_t> rb[NA];
	int64_t rtot[NA];
	for(int a = 0; a < NA; ++a) {
		rb[a].assign((size_t) W, 0);
		rtot[a] = 0;
		for(int src = 0; src < W; ++src) { rb[a][(size_t) src] = all[(size_t) src * NA * W + (size_t) a * W + (size_t) rank]; rtot[a] += rb[a][(size_t) src]; }
	}
	const int64_t m2 = rtot[0] / (ROW * 4);
	int32_t *rows_r = nullptr, *N_r = nullptr;
	uint64_t *seq_r = nullptr;
	uint32_t *ops_r = nullptr;
	if((rc = B.get((size_t) rtot[0] / 4 + ROW, &rows_r)) || (rc = B.get((size_t) rtot[1] / 8 + 2, &seq_r, true)) || (rc = B.get((size_t) rtot[2] / 4 + 1, &N_r)) ||
	   (rc = B.get((size_t) rtot[3] / 4 + 1, &ops_r))) return rc;
	std::vector<char> name_recv((size_t) rtot[4] + 1);
	HIP_TRY(hipStreamSynchronize(s));
	if((rc = kmahip_comm_alltoallv(comm, rows_d, &mine[0], rows_r, rb[0].data(), 1, s)) ||
	   (rc = kmahip_comm_alltoallv(comm, dK.seq, &mine[(size_t) W], seq_r, rb[1].data(), 1, s)) ||
	   (rc = kmahip_comm_alltoallv(comm, dK.N, &mine[(size_t) 2 * W], N_r, rb[2].data(), 1, s)) ||
	   (rc = kmahip_comm_alltoallv(comm, ops_k, &mine[(size_t) 3 * W], ops_r, rb[3].data(), 1, s)) ||
	   (rc = kmahip_comm_alltoallv(comm, name_send.data(), &mine[(size_t) 4 * W], name_recv.data(), rb[4].data(), 0, s))) return rc;
	ms[4] = since(t);

	// the owner's batch: what arrived is in source-rank order = the order of the whole stream
	kmahip_reads dO{};
	kmahip_traces trO;
	memset(&trO, 0, sizeof trO);
	int64_t *fr2 = nullptr, *w_cnt = nullptr, *n_cnt = nullptr, *o_cnt = nullptr, *so2 = nullptr, *no2 = nullptr, *oo2 = nullptr;
	int32_t *len2 = nullptr, *rc2 = nullptr, *tm2 = nullptr, *nh2 = nullptr, *nops2 = nullptr, *st2 = nullptr;
	if((rc = B.get((size_t) m2 + 1, &fr2)) || (rc = B.get((size_t) m2 + 1, &w_cnt)) || (rc = B.get((size_t) m2 + 1, &n_cnt)) || (rc = B.get((size_t) m2 + 1, &o_cnt)) ||
	   (rc = B.get((size_t) m2 + 1, &so2)) || (rc = B.get((size_t) m2 + 1, &no2)) || (rc = B.get((size_t) m2 + 1, &oo2)) || (rc = B.get((size_t) m2 + 1, &len2, true)) ||
	   (rc = B.get((size_t) m2 + 1, &rc2)) || (rc = B.get((size_t) m2 + 1, &tm2)) || (rc = B.get((size_t) m2 + 1, &nh2)) || (rc = B.get((size_t) m2 + 1, &nops2)) ||
	   (rc = B.get((size_t) 10 * m2 + 10, &st2))) return rc;
	hipLaunchKernelGGL(shard_unpack_kernel, dim3((unsigned) ((m2 + 256) / 256)), dim3(256), 0, s, m2, rows_r, fr2, len2, rc2, tm2, nh2, nops2, st2, w_cnt, n_cnt, o_cnt);
	HIP_TRY(hipGetLastError());
	if((rc = scan_i64(B, w_cnt, so2, (size_t) m2 + 1, s)) || (rc = scan_i64(B, n_cnt, no2, (size_t) m2 + 1, s)) || (rc = scan_i64(B, o_cnt, oo2, (size_t) m2 + 1, s))) return rc;
	HIP_TRY(hipStreamSynchronize(s));
	dO.n_reads = m2; dO.seq = seq_r; dO.seq_off = so2; dO.len = len2; dO.N = N_r; dO.N_off = no2; dO.seq_words = rtot[1] / 8; dO.N_total = rtot[2] / 4; dO.max_len = max_len;
	trO.stats = st2; trO.ops_off = oo2; trO.n_ops = nops2; trO.ops = ops_r; trO.ops_cap = rtot[3] / 4;
	// pile-up + consensus of the owned templates
	kmahip_assembly asmb;
	memset(&asmb, 0, sizeof asmb);
	std::vector<int64_t> a_cover(D, 0), a_len(D, 0), a_depth(D, 0), a_asm(D, 0), c_off(D, -1);
	int64_t tbases = 0;
	for(size_t tt = 1; tt < D; ++tt) tbases += db->h_tlen[tt];
	std::vector<char> cons((size_t) (4 * tbases + 4 * (int64_t) D + (1 << 20)));
	asmb.cover = a_cover.data(); asmb.aln_len = a_len.data(); asmb.depth = a_depth.data(); asmb.asm_len = a_asm.data();
	asmb.consensus = cons.data(); asmb.consensus_off = c_off.data(); asmb.consensus_cap = (int64_t) cons.size(); asmb.consensus_used = 0;
	if(m2) {
		kmahip_assemble_opts ao = {chunk, opts->evalue, opts->bcd, order, opts->caller | (opts->ref_fsa == 2 ? 8 : 0) | (opts->write_aln ? 32 : 0), opts->sig90, fr2, opts->support};
		if((rc = kmahip_assemble2_dev(db, ws, &dO, rc2, tm2, &trO, &ao, &asmb))) return rc;
	}
	ms[5] = since(t);

	// the rows of the owned templates: `.res` lines, consensus entries, fragment rows -- parts that rank 0 puts together
	const std::string prefix(out_prefix), part = prefix + ".part" + std::to_string(rank);
	if((rc = kmahip_write_res_fsa(db, (part + ".res").c_str(), (part + ".fsa").c_str(), false, rows, n_rows, owner.data(), rank, a_cover.data(), a_len.data(), a_depth.data(),
	                              cons.data(), c_off.data(), opts->ID_t > 0 ? opts->ID_t : 1.0, opts->Depth_t, opts->ref_fsa, opts->write_aln ? (part + ".aln").c_str() : nullptr))) return rc;
	{
		// the fragment rows are formatted on the host from what arrived (kmahip_frag_write3 with the positions the reads had in the whole stream)
		std::vector<uint64_t> hs((size_t) dO.seq_words + 2, 0);
		std::vector<int64_t> h_so((size_t) m2 + 1, 0), h_no((size_t) m2 + 1, 0), h_fr((size_t) m2 + 1, 0), h_name_off((size_t) m2 + 1, 0);
		std::vector<int32_t> h_len((size_t) m2 + 1, 0), h_N((size_t) dO.N_total + 1, 0), h_rc((size_t) m2 + 1, 0), h_tm((size_t) m2 + 1, 0), h_nh((size_t) m2 + 1, 0), h_st((size_t) 10 * m2 + 10, 0);
		if(dO.seq_words) HIP_TRY(hipMemcpy(hs.data(), seq_r, (size_t) dO.seq_words * 8, hipMemcpyDeviceToHost));
		if(dO.N_total) HIP_TRY(hipMemcpy(h_N.data(), N_r, (size_t) dO.N_total * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(h_so.data(), so2, ((size_t) m2 + 1) * 8, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(h_no.data(), no2, ((size_t) m2 + 1) * 8, hipMemcpyDeviceToHost));
		if(m2) {
			HIP_TRY(hipMemcpy(h_fr.data(), fr2, (size_t) m2 * 8, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(h_len.data(), len2, (size_t) m2 * 4, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(h_rc.data(), rc2, (size_t) m2 * 4, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(h_tm.data(), tm2, (size_t) m2 * 4, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(h_nh.data(), nh2, (size_t) m2 * 4, hipMemcpyDeviceToHost));
			HIP_TRY(hipMemcpy(h_st.data(), st2, (size_t) m2 * 40, hipMemcpyDeviceToHost));
		}
		// names arrive NUL-terminated, back to back
		{
			int64_t at = 0;
			for(int64_t x = 0; x < m2; ++x) { h_name_off[(size_t) x] = at; at += (int64_t) strlen(name_recv.data() + at) + 1; }
			h_name_off[(size_t) m2] = at;
		}
		kmahip_reads hr{};
		hr.n_reads = m2; hr.seq = hs.data(); hr.seq_off = h_so.data(); hr.len = h_len.data(); hr.N = h_N.data(); hr.N_off = h_no.data();
		hr.seq_words = dO.seq_words; hr.N_total = dO.N_total; hr.max_len = max_len;
		int64_t frag_rows = 0;
		if((rc = kmahip_frag_write3((part + ".frag.gz").c_str(), db, &hr, h_rc.data(), h_tm.data(), h_nh.data(), h_st.data(), chunk, order, h_fr.data(),
		                            name_recv.data(), h_name_off.data(), &frag_rows))) return rc;
	}
	ms[6] = since(t);
	if((rc = kmahip_comm_barrier(comm))) return rc;
	if(rank == 0) {
		if((rc = concat_parts(prefix, ".res", W, "#Template\tScore\tExpected\tTemplate_length\tTemplate_Identity\tTemplate_Coverage\tQuery_Identity\tQuery_Coverage\tDepth\tq_value\tp_value\n")) ||
		   (rc = concat_parts(prefix, ".fsa", W, nullptr)) || (rc = concat_parts(prefix, ".frag.gz", W, nullptr)) ||
		   (opts->write_aln && (rc = concat_parts(prefix, ".aln", W, nullptr)))) return rc;
	}
	if((rc = kmahip_comm_barrier(comm))) return rc;
	ms[7] = since(t);
	return KMAHIP_OK;
}

// what follows stage 2 in a sharded run, for a batch of reads (`-1t1`) or of the default mode's records alike: stage 3a, the two
// exchanges around ConClave, the traceback, the positions among the filed fragments of the stream, the gather and the owners' work.
// name_src (host, or NULL): the read of `batch` a record of `d` carries the header of.
static int sharded_after_stage2(kmahip_db *db, kmahip_ws *ws, kmahip_comm *comm, DevBlock &B, const kmahip_read_batch *batch, const kmahip_reads &d, kmahip_cands &c,
                                int64_t total, const int64_t *name_src, const kmahip_params *p, const kmahip_shard_opts *opts, const char *out_prefix, double ms[8],
                                std::chrono::steady_clock::time_point &t) {
	const int W = kmahip_comm_world(comm), rank = kmahip_comm_rank(comm);
	const int64_t n = d.n_reads;
	const size_t D = db->info.DB_size;
	const int64_t mf = opts->max_frag > 0 ? opts->max_frag : 1000000;
	hipStream_t s = 0;
	int rc;
	if(!c.T && (rc = B.get(16, &c.T))) return rc;
	kmahip_hits h;
	uint64_t *AS = nullptr;
	if((rc = B.get((size_t) n + 1, &h.n_hits, true)) || (rc = B.get((size_t) n + 1, &h.best_score, true)) || (rc = B.get((size_t) n + 1, &h.flag, true)) ||
	   (rc = B.get((size_t) n + 1, &h.rc, true)) || (rc = B.get((size_t) total + 1, &h.tmpl, true)) || (rc = B.get((size_t) total + 1, &h.score, true)) ||
	   (rc = B.get((size_t) total + 1, &h.start, true)) || (rc = B.get((size_t) total + 1, &h.end, true)) || (rc = B.get(2 * D, &AS, true))) return rc;
	h.alignment_scores = AS; h.uniq_alignment_scores = AS + D;
	for(;;) {
		if(n && (rc = kmahip_stage3a_se(db, ws, &d, &c, p, &h, s))) return rc;
		HIP_TRY(hipStreamSynchronize(s));
		if(ws_status(ws, nullptr) != 3) break;
		if(!grow_mem_cap(ws)) { kmahip_set_error("seed (MEM) capacity per read/template pair exceeded"); return KMAHIP_EOVERFLOW; }
		HIP_TRY(hipMemsetAsync(AS, 0, 2 * D * 8, s));
	}
	ms[1] = since(t);

	// exchange 1: the two score vectors summed over the shards; ConClave on them; exchange 2: its per-template outputs summed
	if((rc = kmahip_comm_allreduce_u64(comm, AS, 2 * D, s))) return rc;
	kmahip_conclave cc;
	uint64_t *X = nullptr;
	if((rc = B.get((size_t) n + 1, &cc.tmpl, true)) || (rc = B.get((size_t) n + 1, &cc.start, true)) || (rc = B.get((size_t) n + 1, &cc.end, true)) ||
	   (rc = B.get(D, &cc.w_scores, true)) || (rc = B.get(D, &cc.fragment_counts, true)) || (rc = B.get(D, &cc.read_counts, true)) || (rc = B.get(D, &cc.depth, true)) ||
	   (rc = B.get(4 * D, &X))) return rc;
	if(n && (rc = kmahip_conclave_se_dev(db, ws, &d, &c, &h, &cc, s))) return rc;
	hipLaunchKernelGGL(shard_pack_kernel, dim3((unsigned) ((D + 255) / 256)), dim3(256), 0, s, (int64_t) D, cc.w_scores, cc.depth, cc.fragment_counts, cc.read_counts, X);
	HIP_TRY(hipGetLastError());
	if((rc = kmahip_comm_allreduce_u64(comm, X, 4 * D, s))) return rc;
	std::vector<uint64_t> hx(4 * D);
	HIP_TRY(hipMemcpy(hx.data(), X, 4 * D * 8, hipMemcpyDeviceToHost));
	std::vector<kmahip_res_row> rows(D);
	int64_t n_rows = 0;
	if((rc = kmahip_res_rows(db, hx.data(), opts->evalue, p->scoreT, rows.data(), (int64_t) D, &n_rows))) return rc;
	std::vector<uint8_t> ok(D + 8, 0);
	for(int64_t r = 0; r < n_rows; ++r) ok[(size_t) rows[(size_t) r].template_id] = (uint8_t) rows[(size_t) r].significant;
	const uint8_t *d_ok = nullptr;
	if((rc = B.up(ok.data(), D + 8, 0, &d_ok))) return rc;
	ms[2] = since(t);

	// the traceback on the rank's own reads
	kmahip_traces tr;
	if((rc = B.get((size_t) 10 * n + 10, &tr.stats, true)) || (rc = B.get((size_t) n + 1, &tr.ops_off, true)) || (rc = B.get((size_t) n + 1, &tr.n_ops, true))) return rc;
	tr.ops_cap = 6 * n + (1 << 20); tr.ops = nullptr;
	for(int attempt = 0; n; ++attempt) {
		if((rc = B.get((size_t) tr.ops_cap, &tr.ops))) return rc;
		if((rc = kmahip_launch_trace(db, ws, &d, h.rc, cc.tmpl, d_ok, p, &tr, s))) return rc;
		HIP_TRY(hipStreamSynchronize(s));
		unsigned long long used = 0;
		const int st = ws_status(ws, &used);
		if(st == 2 || (int64_t) used > tr.ops_cap) {
			if(attempt >= 2) { kmahip_set_error("alignment run pool: %llu runs needed", used); return KMAHIP_EOVERFLOW; }
			tr.ops_cap = (int64_t) used + (1 << 20);
			continue;
		}
		if(st == 16 && grow_mem_cap(ws)) { --attempt; continue; }
		if(st) { kmahip_set_error("trace stage: a read needs more scratch than the workspace holds (status %d)", st); return KMAHIP_EDEVICE; }
		break;
	}
	if(!tr.ops && (rc = B.get(16, &tr.ops))) return rc;
	ms[3] = since(t);

	// position of every read among the filed fragments of the whole stream: the shards are contiguous in stream order
	int64_t *filed = nullptr, *filed_before = nullptr;
	if((rc = B.get((size_t) n + 1, &filed)) || (rc = B.get((size_t) n + 1, &filed_before))) return rc;
	hipLaunchKernelGGL(shard_filed_kernel, dim3((unsigned) ((n + 256) / 256)), dim3(256), 0, s, n, cc.tmpl, filed);
	HIP_TRY(hipGetLastError());
	if((rc = scan_i64(B, filed, filed_before, (size_t) n + 1, s))) return rc;
	int64_t my_filed = 0;
	HIP_TRY(hipMemcpy(&my_filed, filed_before + n, 8, hipMemcpyDeviceToHost));
	std::vector<int64_t> all_filed((size_t) W);
	if((rc = kmahip_comm_allgather(comm, &my_filed, sizeof my_filed, all_filed.data()))) return rc;
	int64_t rank_base = 0;
	for(int r = 0; r < rank; ++r) rank_base += all_filed[(size_t) r];
	if(n) hipLaunchKernelGGL(shard_add_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, n, filed_before, rank_base);
	HIP_TRY(hipGetLastError());
	return shard_finish(db, ws, comm, B, batch, d, cc.tmpl, h.rc, h.n_hits, filed_before, tr, name_src, &hx[2 * D], rows.data(), n_rows, mf, opts, out_prefix, ms, t);
}

extern "C" int kmahip_run_se_sharded(kmahip_db *db, kmahip_ws *ws, kmahip_comm *comm, const kmahip_read_batch *batch, const kmahip_params *p,
                                     const kmahip_shard_opts *opts, const char *out_prefix, double ms[8]) {
	if(!db || !ws || !batch || !p || !opts || !out_prefix || !ms) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	const kmahip_reads &R = batch->reads;
	const int64_t n = R.n_reads;
	if(n < 0 || R.seq_words < 0 || R.N_total < 0) { kmahip_set_error("negative size"); return KMAHIP_EINVAL; }
	if(n && (!batch->names || !batch->name_off)) { kmahip_set_error("the batch carries no read names"); return KMAHIP_EINVAL; }
	for(int i = 0; i < 8; ++i) ms[i] = 0;
	hipStream_t s = 0;
	DevBlock B;
	int rc;
	auto t = std::chrono::steady_clock::now();
	if((rc = kmahip_db_load_names(db))) return rc;

	// the shard, once
	B.expect((size_t) R.seq_words * 8 + (size_t) R.N_total * 4 + (size_t) n * 320 + (64u << 20));
	kmahip_reads d = R;
	d.q_start = nullptr; d.q_end = nullptr;
	if((rc = B.up(R.seq, (size_t) R.seq_words, 2, &d.seq)) || (rc = B.up(R.seq_off, (size_t) n + 1, 0, &d.seq_off)) || (rc = B.up(R.len, (size_t) n, 1, &d.len)) ||
	   (rc = B.up(R.N, (size_t) R.N_total, 1, &d.N)) || (rc = B.up(R.N_off, (size_t) n + 1, 0, &d.N_off))) return rc;
	HIP_TRY(hipStreamSynchronize(s));
	ms[0] = since(t);

	// stages 2 and 3a on the shard (as kmahip_run_se)
	kmahip_cands c;
	if((rc = B.get((size_t) n + 1, &c.rc_flag)) || (rc = B.get((size_t) n + 1, &c.flag)) || (rc = B.get((size_t) n + 1, &c.T_off, true))) return rc;
	int64_t total = 0;
	c.T_cap = 2 * n + 4096; c.T = nullptr;
	for(int attempt = 0; n > 0; ++attempt) {
		if((rc = B.get((size_t) c.T_cap, &c.T))) return rc;
		if((rc = kmahip_launch_scan_se(db, ws, &d, p, &c, s))) return rc;
		HIP_TRY(hipStreamSynchronize(s));
		if(ws_status(ws, nullptr) == 1) {
			if(attempt >= 4) { kmahip_set_error("internal candidate pool exhausted"); return KMAHIP_EOVERFLOW; }
			ws->pool_scale *= 2; ws->cap_reads = 0;
			continue;
		}
		HIP_TRY(hipMemcpy(&total, c.T_off + n, sizeof total, hipMemcpyDeviceToHost));
		if(total <= c.T_cap) break;
		if(attempt >= 6) { kmahip_set_error("candidate lists keep growing"); return KMAHIP_EOVERFLOW; }
		c.T_cap = total + 1024;
	}
	return sharded_after_stage2(db, ws, comm, B, batch, d, c, total, nullptr, p, opts, out_prefix, ms, t);
}


// ---- the default mode (no -1t1) over read shards: stage 2 (the chain finder) on the rank's reads, its records -- a read, or its pieces,
// with their query bounds -- as a batch in stream order, and from there what a batch of reads goes through (sharded_after_stage2); a
// record carries the header of the read it came from -------------------------------------------------------------------------------
extern "C" int kmahip_run_chain_sharded(kmahip_db *db, kmahip_ws *ws, kmahip_comm *comm, const kmahip_read_batch *batch, const kmahip_params *p,
                                        const kmahip_chain_params *cp, const kmahip_shard_opts *opts, const char *out_prefix, double ms[8]) {
	if(!db || !ws || !comm || !batch || !p || !opts || !out_prefix || !ms) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	const kmahip_reads &R = batch->reads;
	const int64_t n = R.n_reads;
	if(n < 0 || R.seq_words < 0 || R.N_total < 0) { kmahip_set_error("negative size"); return KMAHIP_EINVAL; }
	if(n && (!batch->names || !batch->name_off)) { kmahip_set_error("the batch carries no read names"); return KMAHIP_EINVAL; }
	for(int i = 0; i < 8; ++i) ms[i] = 0;
	hipStream_t s = 0;
	DevBlock B;
	int rc;
	auto t = std::chrono::steady_clock::now();
	if((rc = kmahip_db_load_names(db))) return rc;
	B.expect((size_t) R.seq_words * 16 + (size_t) R.N_total * 8 + (size_t) n * 520 + (64u << 20));
	kmahip_reads dR = R;
	dR.q_start = nullptr; dR.q_end = nullptr;
	if((rc = B.up(R.seq, (size_t) R.seq_words, 2, &dR.seq)) || (rc = B.up(R.seq_off, (size_t) n + 1, 0, &dR.seq_off)) || (rc = B.up(R.len, (size_t) n, 1, &dR.len)) ||
	   (rc = B.up(R.N, (size_t) R.N_total, 1, &dR.N)) || (rc = B.up(R.N_off, (size_t) n + 1, 0, &dR.N_off))) return rc;
	HIP_TRY(hipStreamSynchronize(s));
	ms[0] = since(t);
	ChainRecs CR;
	if((rc = chain_records(db, ws, B, dR, &R, p, cp, CR, [](const char *) {}))) return rc;
	// (every rank's longest read sizes the owners' scratch: the record batch keeps the batch's)
	std::vector<int64_t> h_read((size_t) CR.m + 1, 0);
	if(CR.m) HIP_TRY(hipMemcpy(h_read.data(), CR.o_read, (size_t) CR.m * 8, hipMemcpyDeviceToHost));
	if(!CR.m) {
		// a shard without a record still takes part in every exchange: an empty batch with valid (zero) offsets
		int64_t *zo = nullptr;
		int32_t *zi = nullptr;
		uint64_t *zw = nullptr;
		if((rc = B.get(2, &zo, true)) || (rc = B.get(2, &zi, true)) || (rc = B.get(2, &zw, true))) return rc;
		CR.d = kmahip_reads{};
		CR.d.n_reads = 0; CR.d.seq = zw; CR.d.seq_off = zo; CR.d.len = zi; CR.d.N = zi; CR.d.N_off = zo; CR.d.max_len = R.max_len;
		CR.c.rc_flag = zi; CR.c.flag = zi; CR.c.T_off = zo; CR.c.T = zi; CR.c.T_cap = 1;
	}
	return sharded_after_stage2(db, ws, comm, B, batch, CR.d, CR.c, CR.n_T, h_read.data(), p, opts, out_prefix, ms, t);
}

// ---- `-Mt1 n` over read shards: every rank traces its part of the stream (the traceback is four fifths of the run), two sums make the
// `.res` row the same everywhere, and the kept reads meet at the template's owner -- rank 0 -- with their positions in the whole stream
// for the pile-up in stream order (kmahip_run_mt1; mt1.c:86-500) ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void mt1_kept_kernel(int64_t n, const int32_t *stats, int32_t tmpl, int t_len, int Wl, int32_t *o_tmpl, int32_t *o_nh, int64_t *o_kept,
                                                       unsigned long long *sums) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i > n) return;
	if(i == n) { o_kept[i] = 0; return; }
	const int32_t *st = stats + 10 * i;
	const bool kept = st[3] != 0;
	o_tmpl[i] = kept ? tmpl : 0;
	o_nh[i] = 1;
	o_kept[i] = kept ? 1 : 0;
	// Score of the `.res` row = sum of KMA()'s own scores of the kept reads, without the end bonus the read filter added
	if(kept) { atomicAdd(&sums[0], (unsigned long long) (st[0] - Wl * ((st[1] == 0) + (st[2] == t_len)))); atomicAdd(&sums[1], 1ull); }
}

extern "C" int kmahip_run_mt1_sharded(kmahip_db *db, kmahip_ws *ws, kmahip_comm *comm, const kmahip_read_batch *batch, int32_t tmpl, int one2one,
                                      const kmahip_params *p, const kmahip_shard_opts *opts, const char *out_prefix, double ms[8]) {
	if(!db || !ws || !comm || !batch || !p || !opts || !out_prefix || !ms) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	const int W = kmahip_comm_world(comm), rank = kmahip_comm_rank(comm);
	const kmahip_reads &R = batch->reads;
	const int64_t n = R.n_reads;
	if(n < 0 || R.seq_words < 0 || R.N_total < 0) { kmahip_set_error("negative size"); return KMAHIP_EINVAL; }
	if(n && (!batch->names || !batch->name_off)) { kmahip_set_error("the batch carries no read names"); return KMAHIP_EINVAL; }
	const size_t D = db->info.DB_size;
	if(tmpl < 1 || (size_t) tmpl >= D) { kmahip_set_error("template %d out of range", tmpl); return KMAHIP_EINVAL; }
	const int64_t mf = opts->max_frag > 0 ? opts->max_frag : 1000000;
	for(int i = 0; i < 8; ++i) ms[i] = 0;
	hipStream_t s = 0;
	DevBlock B;
	int rc;
	auto t = std::chrono::steady_clock::now();
	if((rc = kmahip_db_load_names(db))) return rc;
	kmahip_reads d = R;
	d.q_start = nullptr; d.q_end = nullptr;
	if((rc = B.up(R.seq, (size_t) R.seq_words, 2, &d.seq)) || (rc = B.up(R.seq_off, (size_t) n + 1, 0, &d.seq_off)) || (rc = B.up(R.len, (size_t) n, 1, &d.len)) ||
	   (rc = B.up(R.N, (size_t) R.N_total, 1, &d.N)) || (rc = B.up(R.N_off, (size_t) n + 1, 0, &d.N_off))) return rc;
	HIP_TRY(hipStreamSynchronize(s));
	ms[0] = since(t);
	// the traceback on the rank's own reads: strand by anker_rc, runs; the run pool grows on demand
	kmahip_traces tr;
	int32_t *d_rc = nullptr, *d_tmpl = nullptr, *d_nh = nullptr;
	int64_t *kept = nullptr, *kept_before = nullptr;
	unsigned long long *sums = nullptr;
	if((rc = B.get((size_t) 10 * n + 10, &tr.stats, true)) || (rc = B.get((size_t) n + 1, &tr.ops_off, true)) || (rc = B.get((size_t) n + 1, &tr.n_ops, true)) ||
	   (rc = B.get((size_t) n + 1, &d_rc, true)) || (rc = B.get((size_t) n + 1, &d_tmpl, true)) || (rc = B.get((size_t) n + 1, &d_nh, true)) ||
	   (rc = B.get((size_t) n + 1, &kept)) || (rc = B.get((size_t) n + 1, &kept_before)) || (rc = B.get(2, &sums, true))) return rc;
	int64_t total_bases = 0;
	for(int64_t i = 0; i < n; ++i) total_bases += R.len[i];
	tr.ops_cap = total_bases / 3 + 8 * n + (1 << 16); tr.ops = nullptr;
	for(int attempt = 0; n; ++attempt) {
		if((rc = B.get((size_t) tr.ops_cap, &tr.ops))) return rc;
		if((rc = kmahip_launch_longtrace(db, ws, &d, nullptr, tmpl, nullptr, nullptr, one2one, p, &tr, d_rc, s))) return rc;
		unsigned long long used = 0;
		const int st = ws_status(ws, &used);
		if(st == 2 || (int64_t) used > tr.ops_cap) {
			if(attempt >= 2) { kmahip_set_error("alignment run pool: %llu runs needed", used); return KMAHIP_EOVERFLOW; }
			tr.ops_cap = (int64_t) used + (1 << 16);
			continue;
		}
		break;
	}
	if(!tr.ops && (rc = B.get(16, &tr.ops))) return rc;
	ms[3] = since(t);
	// the `.res` row: Score and the number of kept reads summed over the shards; a read's position among the kept reads of the stream
	const int t_len = db->h_tlen[(size_t) tmpl];
	hipLaunchKernelGGL(mt1_kept_kernel, dim3((unsigned) ((n + 256) / 256)), dim3(256), 0, s, n, tr.stats, tmpl, t_len, p->rw.Wl, d_tmpl, d_nh, kept, sums);
	HIP_TRY(hipGetLastError());
	if((rc = scan_i64(B, kept, kept_before, (size_t) n + 1, s))) return rc;
	int64_t my_kept = 0;
	HIP_TRY(hipMemcpy(&my_kept, kept_before + n, 8, hipMemcpyDeviceToHost));
	std::vector<int64_t> all_kept((size_t) W);
	if((rc = kmahip_comm_allgather(comm, &my_kept, sizeof my_kept, all_kept.data()))) return rc;
	int64_t rank_base = 0;
	for(int r = 0; r < rank; ++r) rank_base += all_kept[(size_t) r];
	if(n) hipLaunchKernelGGL(shard_add_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, n, kept_before, rank_base);
	HIP_TRY(hipGetLastError());
	if((rc = kmahip_comm_allreduce_u64(comm, (uint64_t *) sums, 2, s))) return rc;
	unsigned long long h_sums[2] = {0, 0};
	HIP_TRY(hipMemcpy(h_sums, sums, sizeof h_sums, hipMemcpyDeviceToHost));
	const uint64_t score = h_sums[0];
	kmahip_res_row row;
	memset(&row, 0, sizeof row);
	row.template_id = tmpl; row.template_length = t_len; row.score = score; row.expected = 0;
	row.q_value = (double) score; row.p_value = kmahip_p_chisqr((long double) score);
	row.significant = kmahip_cmp(row.p_value <= opts->evalue && score > 0, (double) score >= p->scoreT * t_len);     // mt1.c:419
	std::vector<uint64_t> frag_counts(D, 0);
	frag_counts[(size_t) tmpl] = h_sums[1];
	ms[2] = since(t);
	// a run without a kept read or without a score: nothing is piled up (kmahip_run_mt1) -- the reads then go nowhere
	if(!score && n) HIP_TRY(hipMemsetAsync(d_tmpl, 0, (size_t) n * 4, s));
	return shard_finish(db, ws, comm, B, batch, d, d_tmpl, d_rc, d_nh, kept_before, tr, nullptr, frag_counts.data(), &row, 1, mf, opts, out_prefix, ms, t, 1);
}


// ---- the paired run over read shards: kmahip_run_pe's stages with the exchanges of kmahip_run_se_sharded in between -----------------
static int shard_allreduce(ShardCtx *sc, uint64_t *d_buf, size_t n) { return kmahip_comm_allreduce_u64(sc->comm, d_buf, n, nullptr); }
static uint64_t *shard_frag_counts(ShardCtx *sc, size_t D) { sc->frag_counts.assign(D, 0); return sc->frag_counts.data(); }

// last[4] = {valid, tmpl, start, end}: the first listed hit of this shard's last record with a list; carry = that of the nearest
// earlier shard that has one
static int shard_carry_in(ShardCtx *sc, const int32_t last[4], int32_t carry[3]) {
	const int W = kmahip_comm_world(sc->comm), rank = kmahip_comm_rank(sc->comm);
	std::vector<int32_t> all((size_t) 4 * W);
	int rc = kmahip_comm_allgather(sc->comm, last, 4 * sizeof(int32_t), all.data());
	if(rc) return rc;
	carry[0] = carry[1] = carry[2] = 0;
	for(int r = rank - 1; r >= 0; --r) if(all[(size_t) 4 * r]) { carry[0] = all[(size_t) 4 * r + 1]; carry[1] = all[(size_t) 4 * r + 2]; carry[2] = all[(size_t) 4 * r + 3]; break; }
	return KMAHIP_OK;
}

// The chunks of maxFrag filed fragments close one after the other along the stream (conclave.c:164-196), so the ranks count
// theirs in turn: state = {fragments in the chunk that is open, chunks closed so far}. W rounds of the mailboxes; rank r listens
// for r rounds (receive), then posts its state in the remaining W - r (send).
static int shard_chunk_token(ShardCtx *sc, bool receive, int64_t state[2]) {
	const int W = kmahip_comm_world(sc->comm), rank = kmahip_comm_rank(sc->comm);
	std::vector<int64_t> all((size_t) 3 * W);
	int64_t mine[3] = {receive ? 0 : 1, state[0], state[1]};
	int rc;
	if(receive) {
		state[0] = 0; state[1] = 0;
		for(int round = 0; round < rank; ++round) {
			if((rc = kmahip_comm_allgather(sc->comm, mine, sizeof mine, all.data()))) return rc;
			if(round == rank - 1) {
				if(!all[(size_t) 3 * (rank - 1)]) { kmahip_set_error("chunk token: rank %d had not posted in its round", rank - 1); return KMAHIP_EDEVICE; }
				state[0] = all[(size_t) 3 * (rank - 1) + 1]; state[1] = all[(size_t) 3 * (rank - 1) + 2];
			}
		}
		return KMAHIP_OK;
	}
	for(int round = rank; round < W; ++round) if((rc = kmahip_comm_allgather(sc->comm, mine, sizeof mine, all.data()))) return rc;
	return KMAHIP_OK;
}

static int shard_pe_tail(ShardCtx *sc, kmahip_db *db, kmahip_ws *ws, DevBlock &B, const kmahip_read_batch *batch, const kmahip_reads &dF, const int32_t *f_rc, const int32_t *f_t,
                         const int32_t *f_nh, const int64_t *f_rank, const kmahip_traces &tr, const int64_t *h_src, const kmahip_res_row *rows, int64_t n_rows, int64_t chunk,
                         std::chrono::steady_clock::time_point &t) {
	return shard_finish(db, ws, sc->comm, B, batch, dF, f_t, f_rc, f_nh, f_rank, tr, h_src, sc->frag_counts.data(), rows, n_rows, chunk, sc->opts, sc->out_prefix, sc->ms, t);
}

extern "C" int kmahip_run_pe_sharded(kmahip_db *db, kmahip_ws *ws, kmahip_comm *comm, const kmahip_read_batch *batch, const kmahip_params *p,
                                     const kmahip_shard_opts *opts, const char *out_prefix, double ms[8]) {
	if(!db || !ws || !batch || !p || !opts || !out_prefix || !ms) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	int rc = kmahip_db_load_names(db);
	if(rc) return rc;
	const size_t D = db->info.DB_size;
	for(int i = 0; i < 8; ++i) ms[i] = 0;
	std::vector<kmahip_res_row> rows(D);
	std::vector<int64_t> a0(D), a1(D), a2(D), a3(D);
	kmahip_run run;
	memset(&run, 0, sizeof run);
	run.rows = rows.data(); run.rows_cap = (int64_t) D;
	run.assembly.cover = a0.data(); run.assembly.aln_len = a1.data(); run.assembly.depth = a2.data(); run.assembly.asm_len = a3.data();
	run.caller = opts->caller; run.sig90 = opts->sig90; run.support = opts->support;
	ShardCtx sc{comm, opts, out_prefix, ms, {}};
	rc = run_pe_impl(db, ws, batch, p, opts->evalue, opts->bcd, opts->max_frag, nullptr, &run, &sc, nullptr);
	for(int i = 0; i < 4; ++i) ms[i] = run.ms[i];
	return rc;
}
