// pipeline.hip -- kmahip_run_se: the single-end `-1t1` run of one batch in one call (host buffers in, per-template results
// out), built from the same launchers as the stage-wise entry points; what runKMA does between its input stream and the
// `.res` / consensus output (runkma.c:104-900), minus the files.
#include "kmahip_internal.h"
#include <algorithm>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// Device buffers of one run, carved out of a few large allocations (a hipMalloc per array cost more than ConClave itself:
// thirty of them per run). Everything is released when the run ends.
struct DevBlock {
	std::vector<void *> owned;
	char *slab = nullptr;
	size_t slab_left = 0, slab_bytes = 256u << 20;
	~DevBlock() { for(void *p : owned) (void) hipFree(p); }
	void expect(size_t bytes) { slab_bytes = std::max(slab_bytes, bytes); }
	template <class T> int get(size_t n, T **dst, bool zero = false) {
		const size_t bytes = (((n ? n : 1) * sizeof(T)) + 255) & ~(size_t) 255;
		if(bytes > slab_left) {
			const size_t want = std::max(bytes, slab_bytes);
			void *d = nullptr;
			if(hipMalloc(&d, want) != hipSuccess) {
				// (a smaller slab may still fit)
				if(want == bytes || hipMalloc(&d, bytes) != hipSuccess) { kmahip_set_error("hipMalloc of %zu bytes failed", bytes); return KMAHIP_ENOMEM; }
				owned.push_back(d);
				if(zero && hipMemsetAsync(d, 0, bytes, 0) != hipSuccess) { kmahip_set_error("hipMemset failed"); return KMAHIP_EDEVICE; }
				*dst = (T *) d;
				return KMAHIP_OK;
			}
			owned.push_back(d);
			slab = (char *) d; slab_left = want;
		}
		void *d = slab;
		slab += bytes; slab_left -= bytes;
		if(zero && hipMemsetAsync(d, 0, bytes, 0) != hipSuccess) { kmahip_set_error("hipMemset failed"); return KMAHIP_EDEVICE; }
		*dst = (T *) d;
		return KMAHIP_OK;
	}
	template <class T> int up(const T *src, size_t n, size_t pad, const T **dst) {
		T *d = nullptr;
		int rc = get(n + pad, &d);
		if(rc) return rc;
		if(pad && hipMemsetAsync(d + n, 0, pad * sizeof(T), 0) != hipSuccess) { kmahip_set_error("hipMemset failed"); return KMAHIP_EDEVICE; }
		if(n && hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { kmahip_set_error("hipMemcpy failed"); return KMAHIP_EDEVICE; }
		*dst = d;
		return KMAHIP_OK;
	}
};

double since(std::chrono::steady_clock::time_point &t) {
	const auto now = std::chrono::steady_clock::now();
	const double ms = std::chrono::duration<double, std::milli>(now - t).count();
	t = now;
	return ms;
}

// status word of the workspace after a synchronised stage (and the first counter, the pool / run top)
int ws_status(kmahip_ws *ws, unsigned long long *c0) {
	unsigned long long c[2];
	if(hipMemcpy(c, ws->counters, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return -1;
	if(c[1]) (void) hipMemset(ws->counters + 1, 0, sizeof(unsigned long long));
	if(c0) *c0 = c[0];
	return (int) c[1];
}

}  // namespace

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
	if(n && (rc = kmahip_launch_align_se(db, ws, &d, &c, p, &h, s))) return rc;
	if(dbg) { auto t2 = t; fprintf(stderr, "[kmahip] run_se: stage 3a launched after %.2f ms\n", since(t2)); }
	HIP_TRY(hipStreamSynchronize(s));
	if(ws_status(ws, nullptr) == 3) { kmahip_set_error("seed (MEM) capacity per read/template pair exceeded"); return KMAHIP_EOVERFLOW; }
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
		if(st) { kmahip_set_error("trace stage: a read needs more scratch than the workspace holds (status %d)", st); return KMAHIP_EDEVICE; }
		break;
	}
	out->ms[3] = since(t);

	// stage 3c per template
	if(n) {
		if((rc = kmahip_assemble_dev(db, ws, &d, h.rc, cc.tmpl, &tr, max_frag, bcd, evalue, &out->assembly))) return rc;
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

// ---- paired run: host composition of the stage-wise calls (the record merge is the glue a host program would otherwise write) ----
namespace {

// a batch assembled on the host from reads of another batch
struct HostBatch {
	std::vector<uint64_t> seq;
	std::vector<int64_t> seq_off{0}, N_off{0};
	std::vector<int32_t> len, N;
	std::vector<int64_t> src;          // index of each read in the source batch
	int max_len = 0;
	void add(const kmahip_reads &r, int64_t i) {
		const int L = r.len[i];
		const int64_t w = (L + 31) >> 5;
		seq.insert(seq.end(), r.seq + r.seq_off[i], r.seq + r.seq_off[i] + w);
		seq.push_back(0);
		seq_off.push_back((int64_t) seq.size());
		N.insert(N.end(), r.N + r.N_off[i], r.N + r.N_off[i + 1]);
		N_off.push_back((int64_t) N.size());
		len.push_back(L); src.push_back(i);
		max_len = std::max(max_len, L);
	}
	// the reverse complement of read i (rc_comp, compdna.c:228-256: the bits complemented, an N keeps its place from the other end)
	void add_rc(const kmahip_reads &r, int64_t i) {
		const int L = r.len[i];
		const int64_t w = (L + 31) >> 5;
		const uint64_t *src = r.seq + r.seq_off[i];
		const size_t at = seq.size();
		seq.resize(at + (size_t) w + 1, 0);
		for(int p = 0; p < L; ++p) {
			const int q = L - 1 - p;
			const uint64_t b = 3 - ((src[q >> 5] >> (62 - ((q & 31) << 1))) & 3);
			seq[at + (size_t) (p >> 5)] |= b << (62 - ((p & 31) << 1));
		}
		seq_off.push_back((int64_t) seq.size());
		for(int64_t x = r.N_off[i + 1] - 1; x >= r.N_off[i]; --x) N.push_back(L - 1 - r.N[x]);
		N_off.push_back((int64_t) N.size());
		len.push_back(L); src_read(i);
		max_len = std::max(max_len, L);
	}
	void src_read(int64_t i) { src.push_back(i); }
	kmahip_reads view() {
		if(N.empty()) N.push_back(0);
		kmahip_reads v = {};
		v.n_reads = (int64_t) len.size(); v.seq = seq.data(); v.seq_off = seq_off.data(); v.len = len.data(); v.N = N.data(); v.N_off = N_off.data();
		v.seq_words = (int64_t) seq.size(); v.N_total = N_off.back(); v.max_len = max_len;
		return v;
	}
};

struct HitBuf {
	std::vector<int32_t> n_hits, best, flag, rc, tmpl, score, start, end;
	kmahip_hits view(uint64_t *as, uint64_t *us) {
		kmahip_hits h;
		h.n_hits = n_hits.data(); h.best_score = best.data(); h.flag = flag.data(); h.tmpl = tmpl.data(); h.score = score.data();
		h.start = start.data(); h.end = end.data(); h.alignment_scores = as; h.uniq_alignment_scores = us; h.rc = rc.data();
		return h;
	}
	void size(int64_t n, int64_t cap) {
		n_hits.assign((size_t) n + 1, 0); best.assign((size_t) n + 1, 0); flag.assign((size_t) n + 1, 0); rc.assign((size_t) n + 1, 0);
		tmpl.assign((size_t) cap + 1, 0); score.assign((size_t) cap + 1, 0); start.assign((size_t) cap + 1, 0); end.assign((size_t) cap + 1, 0);
	}
};

}  // namespace

extern "C" int kmahip_run_pe(kmahip_db *db, kmahip_ws *ws, const kmahip_read_batch *batch, const kmahip_params *p, double evalue, int bcd,
                             int64_t max_frag, const char *frag_path, kmahip_run *out) {
	if(!db || !ws || !batch || !p || !out || !out->rows || !out->assembly.cover || !out->assembly.aln_len || !out->assembly.depth || !out->assembly.asm_len || !batch->pair) {
		kmahip_set_error("null argument"); return KMAHIP_EINVAL;
	}
	const kmahip_reads &R = batch->reads;
	const int64_t n = R.n_reads;
	const size_t D = db->info.DB_size;
	for(int i = 0; i < 6; ++i) out->ms[i] = 0;
	out->n_rows = 0;
	auto t = std::chrono::steady_clock::now();
	// units of the stream: a pair (two reads) or a single
	HostBatch PB, SB;
	std::vector<int64_t> unit_first;      // read index of each unit's first read
	std::vector<int32_t> unit_idx;        // >= 0: pair number, < 0: -(single number) - 1
	for(int64_t i = 0; i < n;) {
		unit_first.push_back(i);
		if(batch->pair[i] == 1 && i + 1 < n && batch->pair[i + 1] == 2) { unit_idx.push_back((int32_t) (PB.len.size() / 2)); PB.add(R, i); PB.add(R, i + 1); i += 2; }
		else { unit_idx.push_back(-(int32_t) SB.len.size() - 1); SB.add(R, i); i += 1; }
	}
	const int64_t np = (int64_t) PB.len.size() / 2, ns = (int64_t) SB.len.size();
	std::vector<uint64_t> AS(D, 0), US(D, 0);
	int rc;
	// stages 2 + 3a: pairs
	kmahip_reads pr = PB.view(), sr = SB.view();
	std::vector<int32_t> mate((size_t) 2 * np + 2), prc((size_t) 2 * np + 2), prcf((size_t) 2 * np + 2), pflag((size_t) 2 * np + 2), kind((size_t) np + 1, 0), pT;
	std::vector<int64_t> R_off((size_t) 2 * np + 2, 0);
	HitBuf ph, sh;
	int64_t cap = 8 * np + 1024;
	for(int tries = 0; np > 0; ++tries) {
		pT.assign((size_t) cap + 1, 0);
		ph.size(2 * np, cap);
		std::fill(AS.begin(), AS.end(), 0); std::fill(US.begin(), US.end(), 0);
		kmahip_pe_recs recs = { mate.data(), prc.data(), prcf.data(), pflag.data(), R_off.data(), pT.data(), cap };
		kmahip_hits h = ph.view(AS.data(), US.data());
		rc = kmahip_map_pe(db, ws, &pr, p, &recs, &h, kind.data());
		if(rc == KMAHIP_OK) break;
		if(rc != KMAHIP_EOVERFLOW || tries > 6) return rc;
		cap = std::max<int64_t>(2 * cap, R_off[(size_t) 2 * np] + 16);
	}
	// ... and the single records (their scores add into the same two vectors)
	std::vector<int32_t> srcf((size_t) ns + 1), sflag((size_t) ns + 1), sT;
	std::vector<int64_t> sT_off((size_t) ns + 2, 0);
	std::vector<uint64_t> AS2(D, 0), US2(D, 0);
	cap = 8 * ns + 1024;
	for(int tries = 0; ns > 0; ++tries) {
		sT.assign((size_t) cap + 1, 0);
		sh.size(ns, cap);
		std::fill(AS2.begin(), AS2.end(), 0); std::fill(US2.begin(), US2.end(), 0);
		kmahip_cands cd = { srcf.data(), sflag.data(), sT_off.data(), sT.data(), cap };
		kmahip_hits h = sh.view(AS2.data(), US2.data());
		rc = kmahip_map_se(db, ws, &sr, p, &cd, &h);
		if(rc == KMAHIP_OK) break;
		if(rc != KMAHIP_EOVERFLOW || tries > 6) return rc;
		cap = std::max<int64_t>(2 * cap, sT_off[(size_t) ns] + 16);
	}
	for(size_t i = 0; i < D; ++i) { AS[i] += AS2[i]; US[i] += US2[i]; }
	out->ms[1] = since(t);

	// frag_raw records in stream order (update_Scores_pe / _se, updatescores.c:300-488)
	struct Frag { int64_t read; int32_t flag, rc; };
	std::vector<int32_t> r_n, r_score, r_ql, r_ql2, f_tmpl, f_start, f_end;
	std::vector<int64_t> r_off{0};
	std::vector<std::vector<Frag>> frags;
	auto add = [&](int nh, int score, int ql, int ql2, const HitBuf &src, int64_t o, std::vector<Frag> fr) {
		r_n.push_back(nh); r_score.push_back(score); r_ql.push_back(ql); r_ql2.push_back(ql2);
		for(int x = 0; x < nh; ++x) { f_tmpl.push_back(src.tmpl[(size_t) (o + x)]); f_start.push_back(src.start[(size_t) (o + x)]); f_end.push_back(src.end[(size_t) (o + x)]); }
		r_off.push_back((int64_t) f_tmpl.size());
		frags.push_back(std::move(fr));
	};
	for(size_t u = 0; u < unit_idx.size(); ++u) {
		if(unit_idx[u] < 0) {
			const int64_t j = -(int64_t) unit_idx[u] - 1;
			if(sh.n_hits[(size_t) j] > 0) add(sh.n_hits[(size_t) j], sh.best[(size_t) j], SB.len[(size_t) j], 0, sh, sT_off[(size_t) j], {Frag{unit_first[u], sh.flag[(size_t) j], sh.rc[(size_t) j]}});
			continue;
		}
		const int64_t j = unit_idx[u], r0 = 2 * j, r1 = 2 * j + 1;
		auto ln = [&](int64_t x) { return PB.len[(size_t) (2 * j + mate[(size_t) x])]; };
		auto fg = [&](int64_t x) { return Frag{unit_first[u] + mate[(size_t) x], ph.flag[(size_t) x], ph.rc[(size_t) x] & 1}; };
		const int64_t o = R_off[(size_t) r1];
		const int kd = kind[(size_t) j];
		if(kd == 1) {
			const bool swapped = (ph.rc[(size_t) r1] & 2) != 0;       // the second slot's fragment is written first (alnfrags.c:1807-1812)
			add(ph.n_hits[(size_t) r1], -ph.best[(size_t) r1], swapped ? ln(r1) : ln(r0), swapped ? ln(r0) : ln(r1), ph, o,
			    swapped ? std::vector<Frag>{fg(r1), fg(r0)} : std::vector<Frag>{fg(r0), fg(r1)});
		} else if(kd == 2) {
			const int n0 = ph.n_hits[(size_t) r0], n1 = ph.n_hits[(size_t) r1];
			add(n0, ph.best[(size_t) r0], ln(r0), 0, ph, o, {fg(r0)});
			add(n1, ph.best[(size_t) r1], ln(r1), 0, ph, o + n0, {fg(r1)});
		} else if(kd == 3 || kd == 4) {
			const int64_t x = kd == 3 ? r0 : r1;
			add(ph.n_hits[(size_t) x], ph.best[(size_t) x], ln(x), 0, ph, o, {fg(x)});
		} else {
			for(int64_t x : {r0, r1}) if(mate[(size_t) x] >= 0 && ph.n_hits[(size_t) x] > 0) add(ph.n_hits[(size_t) x], ph.best[(size_t) x], ln(x), 0, ph, R_off[(size_t) x], {fg(x)});
		}
	}
	const int64_t nrec = (int64_t) r_n.size();
	// stage 3b over the records + the `.res` statistics
	std::vector<int32_t> c_tmpl((size_t) nrec + 1, 0), c_start((size_t) nrec + 1, 0), c_end((size_t) nrec + 1, 0);
	std::vector<uint64_t> w(D, 0);
	if(nrec) {
		if(f_tmpl.empty()) { f_tmpl.push_back(0); f_start.push_back(0); f_end.push_back(0); }
		kmahip_hits h;
		memset(&h, 0, sizeof h);
		h.n_hits = r_n.data(); h.best_score = r_score.data(); h.tmpl = f_tmpl.data(); h.start = f_start.data(); h.end = f_end.data();
		h.alignment_scores = AS.data(); h.uniq_alignment_scores = US.data();
		kmahip_conclave cc = { c_tmpl.data(), c_start.data(), c_end.data(), w.data(), nullptr, nullptr, nullptr };
		if((rc = kmahip_conclave_records(db, ws, nrec, r_ql.data(), r_ql2.data(), r_off.data(), &h, &cc))) return rc;
	}
	if((rc = kmahip_res_rows(db, w.data(), evalue, p->scoreT, out->rows, out->rows_cap, &out->n_rows))) return rc;
	std::vector<uint8_t> ok(D + 8, 0);
	for(int64_t r = 0; r < out->n_rows; ++r) ok[(size_t) out->rows[r].template_id] = (uint8_t) out->rows[r].significant;
	out->ms[2] = since(t);

	// the fragments in record order; the first fragment of a record carries the sign of the template (conclave.c:131-146)
	// runConClave closes a chunk of filed fragments when, AFTER a whole record, maxFrag or more have gone in (conclave.c:164-196):
	// a couple that straddles the limit makes a chunk of maxFrag + 1. The chunks are counted here record by record and handed
	// on as positions c (maxFrag + 1) + index, with maxFrag + 1 as the chunk length the pile-up and the writer divide by.
	HostBatch FB;
	std::vector<int32_t> f_rc, f_t, f_nh;
	std::vector<int64_t> f_rank;
	const int64_t mf = max_frag > 0 ? max_frag : 1000000;
	int64_t chunk = 0, in_chunk = 0;
	for(int64_t k = 0; k < nrec; ++k) {
		const int tt = c_tmpl[(size_t) k];
		for(size_t x = 0; x < frags[(size_t) k].size(); ++x) {
			FB.add(R, frags[(size_t) k][x].read);
			f_rc.push_back(frags[(size_t) k][x].rc);
			f_t.push_back(x == 0 ? tt : abs(tt));
			f_nh.push_back(r_n[(size_t) k]);
			f_rank.push_back(tt ? chunk * (mf + 1) + in_chunk + (int64_t) x : 0);
		}
		if(tt) {
			in_chunk += (int64_t) frags[(size_t) k].size();
			if(in_chunk >= mf) { ++chunk; in_chunk = 0; }
		}
	}
	const int64_t nf = (int64_t) FB.len.size();
	kmahip_reads fr = FB.view();
	std::vector<int32_t> stats((size_t) nf * 10 + 10, 0), n_ops((size_t) nf + 1, 0);
	std::vector<int64_t> ops_off((size_t) nf + 1, 0);
	std::vector<uint32_t> ops;
	int64_t ops_cap = 8 * nf + 1024, need = 0;
	kmahip_traces tr;
	memset(&tr, 0, sizeof tr);
	for(int tries = 0; nf > 0; ++tries) {
		ops.assign((size_t) ops_cap + 1, 0);
		tr.stats = stats.data(); tr.ops_off = ops_off.data(); tr.n_ops = n_ops.data(); tr.ops = ops.data(); tr.ops_cap = ops_cap;
		rc = kmahip_align_trace(db, ws, &fr, f_rc.data(), f_t.data(), ok.data(), p, &tr, &need);
		if(rc == KMAHIP_OK) break;
		if(rc != KMAHIP_EOVERFLOW || tries > 3) return rc;
		ops_cap = need + 16;
	}
	out->ms[3] = since(t);
	if(nf > 0) {
		kmahip_assemble_opts ao = {mf + 1, evalue, bcd, 0, 0, 0, f_rank.data()};
		if((rc = kmahip_assemble2(db, ws, &fr, f_rc.data(), f_t.data(), &tr, &ao, &out->assembly))) return rc;
	} else for(size_t i = 0; i < D; ++i) { out->assembly.cover[i] = 0; out->assembly.aln_len[i] = 0; out->assembly.depth[i] = 0; out->assembly.asm_len[i] = 0; }
	out->ms[4] = since(t);
	if(frag_path && nf > 0) {
		if(!batch->names || !batch->name_off) { kmahip_set_error("the batch carries no read names"); return KMAHIP_EINVAL; }
		// names of the fragments, in fragment order
		std::vector<char> names;
		std::vector<int64_t> noff{0};
		for(int64_t i = 0; i < nf; ++i) {
			const char *nm = batch->names + batch->name_off[FB.src[(size_t) i]];
			names.insert(names.end(), nm, nm + strlen(nm) + 1);
			noff.push_back((int64_t) names.size());
		}
		int64_t rows = 0;
		if((rc = kmahip_frag_write3(frag_path, db, &fr, f_rc.data(), f_t.data(), f_nh.data(), stats.data(), mf + 1, 0, f_rank.data(), names.data(), noff.data(), &rows))) return rc;
	}
	out->ms[5] = since(t);
	return KMAHIP_OK;
}

// ---- the default mode (no -1t1): kmahip_scan_chain, then every record through the stages of kmahip_run_se --------------------------
extern "C" int kmahip_run_chain(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const char *names, const int64_t *name_off,
                                const kmahip_params *p, const kmahip_chain_params *cp, double evalue, int bcd, int64_t max_frag,
                                const char *frag_path, kmahip_run *out) {
	if(!db || !ws || !reads || !p || !out || !out->rows || !out->assembly.cover || !out->assembly.aln_len || !out->assembly.depth || !out->assembly.asm_len) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	if(frag_path && (!names || !name_off)) { kmahip_set_error("the fragment file needs the read headers"); return KMAHIP_EINVAL; }
	const int64_t n = reads->n_reads;
	if(n < 0 || reads->seq_words < 0 || reads->N_total < 0) { kmahip_set_error("negative size"); return KMAHIP_EINVAL; }
	const size_t D = db->info.DB_size;
	for(int i = 0; i < 6; ++i) out->ms[i] = 0;
	out->n_rows = 0;
	auto t = std::chrono::steady_clock::now();
	int rc;
	// stage 2: one record per accepted chain
	std::vector<int64_t> r_read, r_Toff;
	std::vector<int32_t> r_flag, r_emit, r_qs, r_qe, r_T;
	kmahip_chain_recs R;
	memset(&R, 0, sizeof R);
	R.rec_cap = 2 * n + 1024; R.T_cap = 16 * n + 4096;
	for(int attempt = 0;; ++attempt) {
		r_read.assign((size_t) R.rec_cap, 0); r_Toff.assign((size_t) R.rec_cap + 1, 0);
		r_flag.assign((size_t) R.rec_cap, 0); r_emit.assign((size_t) R.rec_cap, 0); r_qs.assign((size_t) R.rec_cap, 0); r_qe.assign((size_t) R.rec_cap, 0);
		r_T.assign((size_t) R.T_cap, 0);
		R.read = r_read.data(); R.rc_flag = r_flag.data(); R.emit_rc = r_emit.data(); R.q_start = r_qs.data(); R.q_end = r_qe.data();
		R.T_off = r_Toff.data(); R.T = r_T.data();
		rc = kmahip_scan_chain(db, ws, reads, p, cp, &R);
		if(rc == KMAHIP_EOVERFLOW && attempt < 3 && (R.n_recs > R.rec_cap || R.n_T > R.T_cap)) {
			R.rec_cap = std::max(R.rec_cap, R.n_recs + 16); R.T_cap = std::max(R.T_cap, R.n_T + 16);
			continue;
		}
		if(rc) return rc;
		break;
	}
	const int64_t m = R.n_recs;
	out->ms[1] = since(t);
	const bool dbg = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
	auto t_lap = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) { if(dbg) fprintf(stderr, "[kmahip] run_chain: %s %.1f ms\n", what, since(t_lap)); };
	lap("stage 2 (kmahip_scan_chain)");
	// the records as a batch of their own: the read, or its reverse complement where the record prints that, with its bounds
	// (laid out by a prefix sum, filled by a few threads: two million records one vector push at a time took longer than stage 2)
	std::vector<int64_t> rb_seq_off((size_t) m + 1, 0), rb_N_off((size_t) m + 1, 0);
	std::vector<int32_t> rb_len((size_t) m + 1, 0);
	int rb_max_len = 0;
	for(int64_t x = 0; x < m; ++x) {
		const int64_t r = r_read[(size_t) x];
		const int L = reads->len[r];
		rb_len[(size_t) x] = L;
		rb_seq_off[(size_t) x + 1] = rb_seq_off[(size_t) x] + ((L + 31) >> 5) + 1;
		rb_N_off[(size_t) x + 1] = rb_N_off[(size_t) x] + (reads->N_off[r + 1] - reads->N_off[r]);
		rb_max_len = std::max(rb_max_len, L);
	}
	std::vector<uint64_t> rb_seq((size_t) rb_seq_off[(size_t) m] + 2, 0);
	std::vector<int32_t> rb_N((size_t) rb_N_off[(size_t) m] + 1, 0);
	{
		const int hw = (int) std::thread::hardware_concurrency();
		const int nt = (int) std::max<int64_t>(1, std::min<int64_t>(std::min(16, hw > 0 ? hw : 1), m / 4096));
		auto fill = [&](int w) {
			for(int64_t x = m * w / nt; x < m * (w + 1) / nt; ++x) {
				const int64_t r = r_read[(size_t) x];
				const int L = rb_len[(size_t) x];
				const uint64_t *src = reads->seq + reads->seq_off[r];
				uint64_t *dst = rb_seq.data() + rb_seq_off[(size_t) x];
				int32_t *nd = rb_N.data() + rb_N_off[(size_t) x];
				const int64_t n0 = reads->N_off[r], n1 = reads->N_off[r + 1];
				if(!r_emit[(size_t) x]) {
					memcpy(dst, src, (size_t) ((L + 31) >> 5) * 8);
					for(int64_t y = n0; y < n1; ++y) nd[y - n0] = reads->N[y];
				} else {
					// rc_comp, compdna.c:228-256: the bits complemented, an N keeps its place from the other end
					for(int p = 0; p < L; ++p) {
						const int q = L - 1 - p;
						const uint64_t b = 3 - ((src[q >> 5] >> (62 - ((q & 31) << 1))) & 3);
						dst[p >> 5] |= b << (62 - ((p & 31) << 1));
					}
					for(int64_t y = n1 - 1; y >= n0; --y) nd[n1 - 1 - y] = L - 1 - reads->N[y];
				}
			}
		};
		std::vector<std::thread> pool;
		for(int w = 1; w < nt; ++w) pool.emplace_back(fill, w);
		fill(0);
		for(std::thread &th : pool) th.join();
	}
	kmahip_reads rb = {};
	rb.n_reads = m; rb.seq = rb_seq.data(); rb.seq_off = rb_seq_off.data(); rb.len = rb_len.data(); rb.N = rb_N.data(); rb.N_off = rb_N_off.data();
	rb.seq_words = rb_seq_off[(size_t) m]; rb.N_total = rb_N_off[(size_t) m]; rb.max_len = rb_max_len;
	lap("record batch built");
	DevBlock B;
	B.expect((size_t) rb.seq_words * 8 + (size_t) rb.N_total * 4 + (size_t) m * 300 + (64u << 20));
	kmahip_reads d = rb;
	std::vector<int32_t> zero((size_t) m + 1, 0);
	kmahip_cands c;
	const int32_t *d_rcflag = nullptr, *d_flag = nullptr, *d_T = nullptr;
	const int64_t *d_Toff = nullptr;
	const int64_t total = m ? r_Toff[(size_t) m] : 0;
	if((rc = B.up(rb.seq, (size_t) rb.seq_words, 2, &d.seq)) || (rc = B.up(rb.seq_off, (size_t) m + 1, 0, &d.seq_off)) ||
	   (rc = B.up(rb.len, (size_t) m, 1, &d.len)) || (rc = B.up(rb.N, (size_t) rb.N_total, 1, &d.N)) || (rc = B.up(rb.N_off, (size_t) m + 1, 0, &d.N_off)) ||
	   (rc = B.up(r_qs.data(), (size_t) m, 1, &d.q_start)) || (rc = B.up(r_qe.data(), (size_t) m, 1, &d.q_end)) ||
	   (rc = B.up(r_flag.data(), (size_t) m, 1, &d_rcflag)) || (rc = B.up(zero.data(), (size_t) m, 1, &d_flag)) ||
	   (rc = B.up(r_Toff.data(), (size_t) m + 1, 0, &d_Toff)) || (rc = B.up(r_T.data(), (size_t) total, 1, &d_T))) return rc;
	c.rc_flag = const_cast<int32_t *>(d_rcflag); c.flag = const_cast<int32_t *>(d_flag); c.T_off = const_cast<int64_t *>(d_Toff);
	c.T = const_cast<int32_t *>(d_T); c.T_cap = total + 1;
	HIP_TRY(hipStreamSynchronize(0));
	out->ms[0] = since(t);
	std::vector<int32_t> k_tmpl((size_t) m + 1, 0), k_nh((size_t) m + 1, 0), k_rc((size_t) m + 1, 0), k_stats((size_t) m * 10 + 10, 0);
	PerRead pr = {k_tmpl.data(), k_nh.data(), k_rc.data(), k_stats.data()};
	if(m) {
		if((rc = run_after_stage2(db, ws, B, d, c, total, p, evalue, bcd, max_frag, out, pr, t))) return rc;
	} else {
		std::vector<uint64_t> w(D, 0);
		if((rc = kmahip_res_rows(db, w.data(), evalue, p->scoreT, out->rows, out->rows_cap, &out->n_rows))) return rc;
		for(size_t i = 0; i < D; ++i) { out->assembly.cover[i] = 0; out->assembly.aln_len[i] = 0; out->assembly.depth[i] = 0; out->assembly.asm_len[i] = 0; }
	}
	lap("stages 3a ... pile-up");
	if(frag_path && m) {
		std::vector<char> nm;
		std::vector<int64_t> noff{0};
		for(int64_t x = 0; x < m; ++x) {
			const char *h = names + name_off[r_read[(size_t) x]];
			nm.insert(nm.end(), h, h + strlen(h) + 1);
			noff.push_back((int64_t) nm.size());
		}
		int64_t rows = 0;
		if((rc = kmahip_frag_write(frag_path, db, &rb, k_rc.data(), k_tmpl.data(), k_nh.data(), k_stats.data(), max_frag, nm.data(), noff.data(), &rows))) return rc;
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
	row.significant = ((row.p_value <= aopts->evalue && score > 0) || (double) score >= p->scoreT * t_len) ? 1 : 0;     // mt1.c:434 (cmp = cmp_or)
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
