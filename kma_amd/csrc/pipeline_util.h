// pipeline_util.h -- what the one-call runs (pipeline.hip) and the batched session (session.hip) share: device buffers carved out
// of slabs, the workspace's status word, an exclusive scan, the seed capacity that grows on demand.
#pragma once
#include "kmahip_internal.h"
#include <algorithm>
#include <chrono>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <thread>
#include <vector>

// The large device blocks of a run, kept for the next one (pipeline.hip). A hipMalloc of gigabytes is 0.2 ms as a rule and 0.6-1.4 s now
// and then (tools/malloc_probe.py: one in five of 13 GB) -- the run pool of a million long reads is 13 GB, the reads 2.5 GB, and a run that
// asked for them anew each time took twice as long whenever the driver had to find the memory again.
void *kmahip_devcache_take(size_t bytes, size_t *got);      // a kept block of at least `bytes` (and not over twice that), or NULL
void kmahip_devcache_give(void *p, size_t bytes);           // keep it (or release it, when what is kept would exceed the cache's share)
void kmahip_devcache_flush();                               // release everything kept
// stage 3a of a single-end batch: kmahip_launch_align_se, or under -mem_mode (kmahip_set_mem_mode) the records of runKMA_MEM
int kmahip_stage3a_se(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_cands *cands, const kmahip_params *p, kmahip_hits *out, hipStream_t stream);
int kmahip_mem_mode();

namespace {

// Device buffers of one run, carved out of a few large allocations (a hipMalloc per array cost more than ConClave itself:
// thirty of them per run). When the run ends the small ones are released, the large ones kept for the next run.
struct DevBlock {
	std::vector<std::pair<void *, size_t>> owned;
	char *slab = nullptr;
	size_t slab_left = 0, slab_bytes = 256u << 20;
	~DevBlock() {
		// (hipFree waits for the device before it releases; a block that is kept must not be handed on before that either)
		bool any = false;
		for(auto &b : owned) any = any || b.second >= (64u << 20);
		if(any) (void) hipDeviceSynchronize();
		for(auto &b : owned) { if(b.second >= (64u << 20)) kmahip_devcache_give(b.first, b.second); else (void) hipFree(b.first); }
	}
	void expect(size_t bytes) { slab_bytes = std::max(slab_bytes, bytes); }
	static void *alloc(size_t want, size_t *got) {
		void *d = want >= (64u << 20) ? kmahip_devcache_take(want, got) : nullptr;
		if(d) return d;
		*got = want;
		if(hipMalloc(&d, want) == hipSuccess) return d;
		kmahip_devcache_flush();          // (what is kept may be what is missing)
		return hipMalloc(&d, want) == hipSuccess ? d : nullptr;
	}
	template <class T> int get(size_t n, T **dst, bool zero = false) {
		const size_t bytes = (((n ? n : 1) * sizeof(T)) + 255) & ~(size_t) 255;
		if(bytes > slab_left) {
			const size_t want = std::max(bytes, slab_bytes);
			size_t got = 0;
			void *d = alloc(want, &got);
			if(!d) {
				// (a smaller slab may still fit)
				if(want == bytes || !(d = alloc(bytes, &got))) { kmahip_set_error("hipMalloc of %zu bytes failed", bytes); return KMAHIP_ENOMEM; }
				owned.push_back({d, got});
				if(zero && hipMemsetAsync(d, 0, bytes, 0) != hipSuccess) { kmahip_set_error("hipMemset failed"); return KMAHIP_EDEVICE; }
				*dst = (T *) d;
				return KMAHIP_OK;
			}
			owned.push_back({d, got});
			slab = (char *) d; slab_left = got;
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

// Host buffers for the per-read / per-fragment columns a run brings back: fresh from malloc their pages do not exist yet, and a copy
// that has to fault them in one by one runs at half speed. A thread writes every page once (an atomic OR of zero: it changes nothing,
// whatever has been copied there already) while the device is busy with the stages before the copy.
struct HostCols {
	std::vector<std::pair<char *, size_t>> bufs;
	std::thread th;
	template <class T> T *get(size_t n) {
		T *p = (T *) malloc((n ? n : 1) * sizeof(T));
		if(p) bufs.push_back({(char *) p, n * sizeof(T)});
		return p;
	}
	void start() {
		th = std::thread([this] { for(auto &b : bufs) for(size_t i = 0; i < b.second; i += 4096) __atomic_fetch_or(&b.first[i], 0, __ATOMIC_RELAXED); });
	}
	void wait() { if(th.joinable()) th.join(); }
	~HostCols() { wait(); for(auto &b : bufs) free(b.first); }
};

// status word of the workspace after a synchronised stage (and the first counter, the pool / run top)
int ws_status(kmahip_ws *ws, unsigned long long *c0) {
	unsigned long long c[2];
	if(hipMemcpy(c, ws->counters, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return -1;
	if(c[1]) (void) hipMemset(ws->counters + 1, 0, sizeof(unsigned long long));
	if(c0) *c0 = c[0];
	return (int) c[1];
}

// a read full of repeats can carry more MEMs against a template than the scratch holds slots for (64 for reads up to 1 kb;
// status 3 from stage 3a, 16 from the traceback): the capacity goes up fourfold and the stage is run again
bool grow_mem_cap(kmahip_ws *ws) {
	const int cur = ws->mem_scale > 0 ? ws->mem_scale : 1;
	if(cur >= 64) return false;
	ws->mem_scale = cur * 4;
	if(getenv("KMAHIP_DEBUG_TIMING")) fprintf(stderr, "[kmahip] seed (MEM) capacity per read and template raised to %d x the usual\n", ws->mem_scale);
	return true;
}


int scan_i64(DevBlock &B, const int64_t *in, int64_t *out, size_t n, hipStream_t s) {
	size_t tmp_bytes = 0;
	if(rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, (int64_t) 0, n, rocprim::plus<int64_t>(), s) != hipSuccess) {
		kmahip_set_error("rocprim::exclusive_scan (size query) failed"); return KMAHIP_EDEVICE;
	}
	char *tmp = nullptr;
	int rc = B.get(tmp_bytes, &tmp);
	if(rc) return rc;
	if(rocprim::exclusive_scan(tmp, tmp_bytes, in, out, (int64_t) 0, n, rocprim::plus<int64_t>(), s) != hipSuccess) {
		kmahip_set_error("rocprim::exclusive_scan failed"); return KMAHIP_EDEVICE;
	}
	return KMAHIP_OK;
}


}  // namespace
