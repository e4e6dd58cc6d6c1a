// db.hip -- host side of libkmahip: reads the reference's on-disk index
// (<prefix>.comp.b/.length.b/.seq.b; written by `kma index`,
// hashmapkma.c:722-775) and lays the database out in HBM for the gfx950 kernels.
#include "kmahip_internal.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <thread>
#include <rocprim/rocprim.hpp>

static thread_local char g_err[512] = "";

void kmahip_set_error(const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
}

extern "C" const char *kmahip_last_error(void) { return g_err; }

extern "C" void kmahip_default_params(kmahip_params *p) {
	// kma.c:327-336 defaults; MM = (Ts + Tv - 1) / 2 (kma.c:1308); d: kma.c:1309-1328
	const int Ts = -2, Tv = -2;
	memset(p, 0, sizeof *p);
	p->rw.M = 1; p->rw.U = -1; p->rw.W1 = -3; p->rw.Wl = -6; p->rw.Mn = 0; p->rw.PE = 7;
	p->rw.MM = (Ts + Tv - 1) / 2;
	for(int i = 0; i < 4; ++i) {
		for(int j = 0; j < 4; ++j) p->rw.d[i][j] = Tv;
		p->rw.d[i][4] = p->rw.Mn;
		p->rw.d[i][i ^ 2] = Ts;
		p->rw.d[i][i] = p->rw.M;
	}
	for(int j = 0; j < 5; ++j) p->rw.d[4][j] = p->rw.Mn;
	p->rw.d[4][4] = 0;
	p->exhaustive = 0; p->minlen = 16; p->mq = 0;
	p->scoreT = 0.5; p->mrc = 0.0; p->minFrac = 1.0; p->ts = 0; p->apm = 0;
}

static int g_device = 0;

extern "C" int kmahip_init(int device) {
	int n = 0;
	HIP_TRY(hipGetDeviceCount(&n));
	if(device < 0 || device >= n) { kmahip_set_error("device %d out of range (%d visible)", device, n); return KMAHIP_EINVAL; }
	HIP_TRY(hipSetDevice(device));
	g_device = device;
	return KMAHIP_OK;
}

template <class T>
static int upload(kmahip_db *db, const T *src, size_t count, const T **dst) {
	void *d = nullptr;
	size_t bytes = (count ? count : 1) * sizeof(T);
	HIP_TRY(hipMalloc(&d, bytes));
	db->allocs.push_back(d);
	if(count && src) HIP_TRY(hipMemcpy(d, src, count * sizeof(T), hipMemcpyHostToDevice));      // (src == NULL: room only)
	*dst = (const T *) d;
	db->info.total_bytes += bytes;
	return KMAHIP_OK;
}

static bool read_exact(FILE *f, void *dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes; }

static inline uint32_t home_bucket(uint32_t key, uint32_t nb_log2) {
	return (uint32_t) (key * 0x9E3779B1u) >> (32 - nb_log2);
}

// ---- the walkable template store on the device: vs_id[g] = value-list offset of the k-mer that starts at position g of `cat`, and every
// slot of the probe table re-pointed from its list offset to the FIRST position of its k-mer (what a serial pass over the templates
// gives; here an atomicMin per k-mer start) ---------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ int find_slot(const uint2 *slots, uint32_t nb_log2, uint32_t km, int64_t *si_out) {
	const uint32_t nbm = (1u << nb_log2) - 1u;
	uint32_t b = (km * 0x9E3779B1u) >> (32u - nb_log2);
	for(;;) {
		const uint2 *sl = slots + (size_t) b * KMAHIP_BUCKET_SLOTS;
		for(int j = 0; j < KMAHIP_BUCKET_SLOTS; ++j) {
			if(sl[j].y == KMAHIP_EMPTY_VI) return 0;            // k-mer not in the index
			if(sl[j].x == km) { *si_out = (int64_t) b * KMAHIP_BUCKET_SLOTS + j; return 1; }
		}
		b = (b + 1u) & nbm;
	}
}

// the probe table and the presence bits from the index's (k-mer, list offset) pairs: a lane per k-mer takes the first free slot at
// or behind its home bucket with a 64-bit compare-and-swap. Slots of a bucket are tried in order, so a bucket fills front to back
// and "last slot empty" keeps meaning "a free slot" for the readers; which of several k-mers of one home bucket ends up in the
// next bucket depends on who came first -- a lookup finds each of them either way.
__global__ __launch_bounds__(256) void table_fill_kernel(unsigned long long *slots, int64_t n_slots) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i < n_slots) slots[i] = (unsigned long long) KMAHIP_EMPTY_VI << 32;
}

__global__ __launch_bounds__(256) void table_insert_kernel(const uint32_t *keys, const uint32_t *vidx, int64_t n, unsigned long long *slots, uint32_t nb_log2,
                                                           uint32_t *kbits, uint32_t kbits_shift) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n) return;
	const uint32_t key = keys[i], nbm = (1u << nb_log2) - 1u;
	const unsigned long long empty = (unsigned long long) KMAHIP_EMPTY_VI << 32, mine = ((unsigned long long) vidx[i] << 32) | key;
	if(kbits) {
		const uint32_t h = (key * KMAHIP_KBITS_MUL) >> kbits_shift;
		atomicOr(&kbits[h >> 5], 1u << (h & 31));
	}
	uint32_t b = (key * 0x9E3779B1u) >> (32u - nb_log2);
	for(;;) {
		unsigned long long *s = slots + (size_t) b * KMAHIP_BUCKET_SLOTS;
		for(int j = 0; j < KMAHIP_BUCKET_SLOTS; ++j) if(atomicCAS(&s[j], empty, mine) == empty) return;
		b = (b + 1u) & nbm;
	}
}

// one workgroup per template; slots[].y still holds the list offsets
__global__ __launch_bounds__(256) void walk_vsid_kernel(const uint2 *slots, uint32_t nb_log2, const uint64_t *cat, const int64_t *cat_off, const int32_t *tlen,
                                                        int k, uint32_t *vs_id, uint32_t *first) {
	const uint32_t t = blockIdx.x + 1;
	const int64_t g0 = cat_off[t];
	const int tl = tlen[t];
	// (blockIdx.y: a long template -- a genome -- is shared by several workgroups)
	for(int i = threadIdx.x + blockIdx.y * blockDim.x; i + k <= tl; i += blockDim.x * gridDim.y) {
		const int64_t g = g0 + i;
		const int ip = (int) (g & 31) << 1;
		uint64_t x = cat[g >> 5] << ip;
		if(ip) x |= cat[(g >> 5) + 1] >> (64 - ip);
		const uint32_t km = (uint32_t) (x >> (64 - 2 * k));
		int64_t si;
		if(!find_slot(slots, nb_log2, km, &si)) continue;
		vs_id[g] = slots[si].y;
		atomicMin(&first[si], (uint32_t) g);
	}
}

__global__ __launch_bounds__(256) void walk_repoint_kernel(uint2 *slots, int64_t n_slots, const uint32_t *first, unsigned long long *unplaced) {
	const int64_t si = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(si >= n_slots || slots[si].y == KMAHIP_EMPTY_VI) return;
	if(first[si] == 0xFFFFFFFFu) { atomicAdd(unplaced, 1ull); return; }
	slots[si].y = first[si];
}


// ---- the per-template position index on the device: every k-mer start as (template, k-mer) -> 1-based position, sorted (a stable
// radix sort keeps the positions of a repeated k-mer ascending), one lane per distinct pair writes its entry -- the position, or a
// reference to "count, positions ..." in the duplicate list, laid out in (template, k-mer) order by a prefix sum -- into the
// template's linear-probing table with a compare-and-swap -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void tpos_keys_kernel(const uint64_t *cat, const int64_t *cat_off, const int32_t *tlen, int k, unsigned long long *keys, int32_t *vals) {
	const uint32_t t = blockIdx.x + 1;
	const int64_t g0 = cat_off[t];
	const int tl = tlen[t];
	for(int i = threadIdx.x + blockIdx.y * blockDim.x; i < tl; i += blockDim.x * gridDim.y) {
		const int64_t g = g0 + i;
		unsigned long long key = ~0ull;
		int32_t val = 0;
		if(i + k <= tl) {
			const int ip = (int) (g & 31) << 1;
			uint64_t x = cat[g >> 5] << ip;
			if(ip) x |= cat[(g >> 5) + 1] >> (64 - ip);
			const uint32_t km = (uint32_t) (x >> (64 - 2 * k));
			if(km) { key = ((unsigned long long) t << 32) | km; val = i + 1; }      // (the poly-A k-mer is never indexed, hashmapcci.c:414-417)
		}
		keys[g] = key; vals[g] = val;
	}
}

__global__ __launch_bounds__(256) void tpos_runs_kernel(const unsigned long long *ks, int64_t n, int64_t *need) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i > n) return;
	int64_t v = 0;
	if(i < n) {
		const unsigned long long key = ks[i];
		if(key != ~0ull && (i == 0 || ks[i - 1] != key)) {
			int64_t len = 1;
			while(i + len < n && ks[i + len] == key) ++len;
			if(len > 1) v = len + 1;
		}
	}
	need[i] = v;
}

__global__ __launch_bounds__(256) void tpos_insert_kernel(const unsigned long long *ks, const int32_t *vs, int64_t n, const int64_t *need, const int64_t *doff,
                                                          const int64_t *poff, const uint32_t *pshift, unsigned long long *slots, int32_t *dups) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n) return;
	const unsigned long long key = ks[i];
	if(key == ~0ull || (i > 0 && ks[i - 1] == key)) return;
	const uint32_t t = (uint32_t) (key >> 32), km = (uint32_t) key;
	int32_t val;
	if(need[i] == 0) val = vs[i];
	else {
		const int64_t G = 1 + doff[i], len = need[i] - 1;
		dups[G] = (int32_t) len;
		for(int64_t c = 0; c < len; ++c) dups[G + 1 + c] = vs[i + c];
		val = -(int32_t) (G + 1);
	}
	const uint32_t sh = pshift[t], msk = (uint32_t) ((1ull << (32 - sh)) - 1);
	unsigned long long *tab = slots + poff[t];
	const unsigned long long mine = ((unsigned long long) (uint32_t) val << 32) | km;
	uint32_t sl = (km * 0x9E3779B1u) >> sh;
	while(atomicCAS(&tab[sl], 0ull, mine) != 0ull) sl = (sl + 1) & msk;
}

}  // namespace

extern "C" int kmahip_db_open(const char *prefix, kmahip_db **out) {
	if(!prefix || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	*out = nullptr;
	std::string base(prefix);
	// KMAHIP_DEBUG_TIMING: where the time of opening an index goes
	const bool dbg = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	auto stamp = [&](const char *what) {
		if(!dbg) return;
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "[kmahip] db_open: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
		t_last = now;
	};
	FILE *f = fopen((base + ".comp.b").c_str(), "rb");
	if(!f) { kmahip_set_error("cannot open %s.comp.b", prefix); return KMAHIP_EIO; }
	uint32_t h32[3];
	uint64_t h64[5];
	if(!read_exact(f, h32, 12) || !read_exact(f, h64, 40)) { fclose(f); kmahip_set_error("short header in %s.comp.b", prefix); return KMAHIP_EIO; }
	const uint32_t DB_size = h32[0], mlen = h32[1];
	const uint64_t size = h64[1], n_hdr = h64[2], v_index = h64[3];
	// `kma index -Sparse`: only the k-mers behind a prefix are stored and the reference maps with save_kmers_sparse, without
	// stage 3 (kma.c:1499-1501); the -1t1 scan over such a table would be wrong without any error
	if(h32[2] != 0 || h64[0] != 0) { fclose(f); kmahip_set_error("sparse index (prefix length %u) not supported", h32[2]); return KMAHIP_EFORMAT; }
	if(mlen == 0 || mlen > 16) { fclose(f); kmahip_set_error("k-mer length %u needs 64-bit keys: not supported", mlen); return KMAHIP_EFORMAT; }
	const uint64_t kmask = (1ull << (2 * mlen)) - 1;
	// A direct-address index (`kma index -ME`, or any index whose table grew to 4^k slots, hashmap.c:204-209): exist[] holds one
	// entry per possible k-mer, the offset of its value list or 1 for "none" (megaMap_getGlobal, hashmapkma.c:264-273), and
	// there are no key / value-index arrays. The k-mers present are read out of it in ascending order; from there on the
	// index is built like any other (the probe table below is what the kernels look k-mers up in).
	const bool mega = size - 1 == kmask;
	uint64_t n = n_hdr;
	if(size < n_hdr || n_hdr > 0xFFFFFFFFull || v_index >= 0xFFFFFFFFull || (n_hdr == 0 && !mega)) { fclose(f); kmahip_set_error("index too large or old format"); return KMAHIP_EFORMAT; }
	const bool u16 = DB_size < 65535; // hashmapkma.c:340-348
	// 8 zero elements of slack: the scan kernel fetches a list head as count + 7 ids
	const size_t vbytes = v_index * (u16 ? 2 : 4);
	std::vector<uint8_t> values(vbytes + 8 * 4, 0);
	std::vector<uint32_t> keys, vidx;
	uint32_t tail[2] = {mlen, 0};
	bool ok = true;
	if(mega) {
		std::vector<uint32_t> chunk(1u << 22);
		for(uint64_t at = 0; ok && at < size; at += chunk.size()) {
			const size_t m = (size_t) std::min<uint64_t>(chunk.size(), size - at);
			ok = read_exact(f, chunk.data(), m * 4);
			for(size_t i = 0; ok && i < m; ++i) if(chunk[i] != 1u) { keys.push_back((uint32_t) (at + i)); vidx.push_back(chunk[i]); }
		}
		n = keys.size();
		keys.push_back(0);
		if(ok && (n == 0 || n > 0xFFFFFFFFull)) { fclose(f); kmahip_set_error("direct-address index holds no k-mers"); return KMAHIP_EFORMAT; }
		ok = ok && read_exact(f, values.data(), vbytes);
	} else {
		// exist[] is only the reference's bucket directory: skipped
		if(fseek(f, (long) (size * 4), SEEK_CUR)) { fclose(f); return KMAHIP_EIO; }
		keys.resize(n + 1); vidx.resize(n);
		ok = read_exact(f, values.data(), vbytes) && read_exact(f, keys.data(), (n + 1) * 4) && read_exact(f, vidx.data(), n * 4);
	}
	if(ok && read_exact(f, &tail[0], 4)) ok = read_exact(f, &tail[1], 4);
	fclose(f);
	if(!ok) { kmahip_set_error("truncated %s.comp.b", prefix); return KMAHIP_EIO; }
	stamp("read .comp.b");
	if(tail[1] != 0) { kmahip_set_error("minimizer / homopolymer-compressed index (flag %u) not supported", tail[1]); return KMAHIP_EFORMAT; }
	if(tail[0] != mlen) { kmahip_set_error("kmersize %u != mlen %u not supported", tail[0], mlen); return KMAHIP_EFORMAT; }

	kmahip_db *db = new kmahip_db();
	db->prefix = base;
	memset(&db->info, 0, sizeof db->info);
	db->device = g_device;
	db->info.DB_size = DB_size; db->info.kmersize = tail[0]; db->info.n_kmers = n; db->info.n_values = v_index;

	// probe table: >= 2n slots -> load factor in (0.25, 0.5]
	uint32_t nb_log2 = 4;
	while(((uint64_t) KMAHIP_BUCKET_SLOTS << nb_log2) < 2 * n) ++nb_log2;
	if(nb_log2 > 31) { delete db; kmahip_set_error("index too large"); return KMAHIP_EFORMAT; }
	const uint64_t nb = 1ull << nb_log2;
	const size_t n_slots = (size_t) nb * KMAHIP_BUCKET_SLOTS;
	db->info.hash_bytes = n_slots * sizeof(uint2);
	// presence bits: ~7 bits per k-mer (13 % false positives) as long as that fits 2 MiB
	uint32_t kbits_log2 = 0;
	if(n > 0 && 8 * n < (1ull << 25) && !getenv("KMAHIP_NO_KBITS")) {        // (the switch exists for the test that compares both paths)
		kbits_log2 = 16;
		while((2ull << kbits_log2) <= 8 * n) ++kbits_log2;
	} else if(n > 0 && n <= (1ull << 23) && !getenv("KMAHIP_NO_KBITS")) {
		// up to 8 M k-mers (a bacterial genome): the 2 MiB that stay in L2, at 2-4 bits per k-mer (26 % false positives at 5 M, 39 % at
		// 8 M) -- still worth it where nine lookups in ten miss: the seeding of long reads against one genome (longtrace.hip)
		kbits_log2 = 24;
	}

	int rc;
	DevDB &d = db->dev;
	memset(&d, 0, sizeof d);
	d.DB_size = DB_size; d.kmersize = tail[0]; d.mlen = mlen; d.nb_log2 = nb_log2; d.values_u16 = u16;
	// the probe table and the presence bits, filled on the device from the k-mers and their list offsets
	uint2 *d_slots = nullptr;
	{
		uint32_t *d_kbits = nullptr;
		const uint32_t *d_keys = nullptr, *d_vidx = nullptr;
		if((rc = upload(db, (const uint2 *) nullptr, n_slots, (const uint2 **) &d_slots))) { kmahip_db_close(db); return rc; }
		if(kbits_log2 && (rc = upload(db, (const uint32_t *) nullptr, (size_t) 1 << (kbits_log2 - 5), (const uint32_t **) &d_kbits))) { kmahip_db_close(db); return rc; }
		bool bad = hipMalloc((void **) &d_keys, (size_t) (n + 1) * 4) != hipSuccess || hipMalloc((void **) &d_vidx, (size_t) (n + 1) * 4) != hipSuccess;
		bad = bad || hipMemcpy((void *) d_keys, keys.data(), (size_t) n * 4, hipMemcpyHostToDevice) != hipSuccess ||
		      hipMemcpy((void *) d_vidx, vidx.data(), (size_t) n * 4, hipMemcpyHostToDevice) != hipSuccess;
		if(!bad && d_kbits) bad = hipMemsetAsync(d_kbits, 0, (size_t) 4 << (kbits_log2 - 5), 0) != hipSuccess;
		if(!bad) {
			hipLaunchKernelGGL(table_fill_kernel, dim3((unsigned) ((n_slots + 255) / 256)), dim3(256), 0, 0, (unsigned long long *) d_slots, (int64_t) n_slots);
			if(n) hipLaunchKernelGGL(table_insert_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, d_keys, d_vidx, (int64_t) n, (unsigned long long *) d_slots,
			                         nb_log2, d_kbits, 32 - kbits_log2);
			bad = hipDeviceSynchronize() != hipSuccess;
		}
		(void) hipFree((void *) d_keys); (void) hipFree((void *) d_vidx);
		if(bad) { kmahip_db_close(db); kmahip_set_error("building the probe table failed: %s", hipGetErrorString(hipGetLastError())); return KMAHIP_EDEVICE; }
		d.slots = d_slots;
		if(d_kbits) { d.kbits = d_kbits; d.kbits_shift = 32 - kbits_log2; }
	}
	if(u16) rc = upload(db, (const uint16_t *) values.data(), (size_t) v_index + 8, &d.values16);
	else rc = upload(db, (const uint32_t *) values.data(), (size_t) v_index + 8, &d.values32);
	if(rc) { kmahip_db_close(db); return rc; }

	// template lengths + 2-bit template store
	f = fopen((base + ".length.b").c_str(), "rb");
	if(!f) { kmahip_db_close(db); kmahip_set_error("cannot open %s.length.b", prefix); return KMAHIP_EIO; }
	{
		int32_t cnt = 0;
		if(!read_exact(f, &cnt, 4) || (uint32_t) cnt != DB_size) { fclose(f); kmahip_db_close(db); kmahip_set_error("bad %s.length.b", prefix); return KMAHIP_EIO; }
		db->h_tlen.resize(DB_size);
		if(!read_exact(f, db->h_tlen.data(), (size_t) DB_size * 4)) { fclose(f); kmahip_db_close(db); return KMAHIP_EIO; }
		fclose(f);
		std::vector<int64_t> off(DB_size + 1);
		off[0] = 0; if(DB_size > 0) off[1] = 0;
		for(uint32_t i = 2; i <= DB_size; ++i) off[i] = off[i - 1] + (db->h_tlen[i - 1] >> 5) + 1; // runkma.c:214-220
		std::vector<uint64_t> tseq((size_t) off[DB_size] + 2, 0);
		f = fopen((base + ".seq.b").c_str(), "rb");
		if(!f || !read_exact(f, tseq.data(), (size_t) off[DB_size] * 8)) { if(f) fclose(f); kmahip_db_close(db); kmahip_set_error("bad %s.seq.b", prefix); return KMAHIP_EIO; }
		fclose(f);
		db->info.tseq_words = off[DB_size];
		if((rc = upload(db, db->h_tlen.data(), db->h_tlen.size(), &d.tlen)) ||
		   (rc = upload(db, tseq.data(), tseq.size(), &d.tseq)) ||
		   (rc = upload(db, off.data(), off.size(), &d.tseq_off))) { kmahip_db_close(db); return rc; }

		stamp("probe table, presence bits, value lists, template store (built + uploaded)");
		int max_tlen = 0;
		for(uint32_t t = 1; t < DB_size; ++t) max_tlen = std::max(max_tlen, db->h_tlen[t]);
		const unsigned grid_y = (unsigned) std::max(1, std::min(1024, max_tlen / 4096));      // workgroups per template in the per-position kernels
		// Concatenated template store + per-position value-list offsets: a read that matches a template keeps
		// matching it, so after one hash hit the scan kernel walks along `cat` (sequential 4-byte reads of
		// vs_id) instead of probing the table for every k-mer start.
		{
			const int kk = (int) tail[0];
			int64_t total = 0;
			for(uint32_t t = 1; t < DB_size; ++t) total += db->h_tlen[t];
			if(total + 64 >= 0xFFFFFFFFll) { kmahip_db_close(db); kmahip_set_error("template store too large for 32-bit positions"); return KMAHIP_EFORMAT; }
			std::vector<uint64_t> cat((size_t) (total >> 5) + 4, 0);
			int64_t g0 = 0;
			db->h_cat_off.assign((size_t) DB_size + 1, 0);
			for(uint32_t t = 1; t < DB_size; ++t) {
				const int tl = db->h_tlen[t];
				const uint64_t *ts = tseq.data() + off[t];
				// the template's words shifted into place, 32 bases at a time (what lies behind its last base is masked off)
				const int sh = (int) (g0 & 31) << 1;
				uint64_t *dst = cat.data() + (g0 >> 5);
				const int words = (tl + 31) >> 5;
				for(int w = 0; w < words; ++w) {
					uint64_t v = ts[w];
					const int have = tl - 32 * w;
					if(have < 32) v &= ~0ull << (64 - 2 * have);
					dst[w] |= v >> sh;
					if(sh) dst[w + 1] |= v << (64 - sh);
				}
				g0 += tl;
				db->h_cat_off[t + 1] = g0;
			}
			if(DB_size > 1) db->h_cat_off[1] = 0;
			uint32_t *d_vsid = nullptr, *d_first = nullptr;
			unsigned long long *d_unplaced = nullptr, unplaced = 0;
			if((rc = upload(db, cat.data(), cat.size(), &d.cat)) ||
			   (rc = upload(db, db->h_cat_off.data(), db->h_cat_off.size(), &d.cat_off)) ||
			   (rc = upload(db, (const uint32_t *) nullptr, (size_t) total + 64, (const uint32_t **) &d_vsid))) { kmahip_db_close(db); return rc; }
			d.vs_id = d_vsid;
			bool bad = hipMalloc((void **) &d_first, n_slots * 4 + 8) != hipSuccess;
			if(!bad) {
				d_unplaced = (unsigned long long *) (d_first + n_slots);
				bad = hipMemsetAsync(d_first, 0xFF, n_slots * 4, 0) != hipSuccess || hipMemsetAsync(d_unplaced, 0, 8, 0) != hipSuccess ||
				      hipMemsetAsync(d_vsid, 0xFF, ((size_t) total + 64) * 4, 0) != hipSuccess;
				if(!bad && DB_size > 1) {
					hipLaunchKernelGGL(walk_vsid_kernel, dim3(DB_size - 1, grid_y), dim3(256), 0, 0, d_slots, nb_log2, d.cat, d.cat_off, d.tlen, kk, d_vsid, d_first);
					hipLaunchKernelGGL(walk_repoint_kernel, dim3((unsigned) ((n_slots + 255) / 256)), dim3(256), 0, 0, d_slots, (int64_t) n_slots, d_first, d_unplaced);
				}
				bad = bad || hipMemcpy(&unplaced, d_unplaced, 8, hipMemcpyDeviceToHost) != hipSuccess;
				(void) hipFree(d_first);
			}
			if(bad) { kmahip_db_close(db); kmahip_set_error("building the walkable template store failed: %s", hipGetErrorString(hipGetLastError())); return KMAHIP_EDEVICE; }
			if(unplaced) { kmahip_db_close(db); kmahip_set_error("%llu index k-mers do not occur in %s.seq.b", unplaced, prefix); return KMAHIP_EFORMAT; }
		}

		stamp("walkable template store (cat / vs_id)");
		// per-template k-mer position index
		const int k = (int) tail[0];
		std::vector<int64_t> poff(DB_size + 1, 0);
		std::vector<uint32_t> pshift(DB_size, 31);
		for(uint32_t t = 1; t < DB_size; ++t) {
			const int64_t nk = std::max<int64_t>(0, (int64_t) db->h_tlen[t] - k + 1);
			uint32_t lg = 4;
			// load <= 1/3: the kernels that look k-mers up wait for the longest of their probe chains (the long-read seeding 256 lookups side
			// by side: fifteen round trips at 2/3, two or three now; the short-read seeding 2.96 -> 2.10 ms per 10 M reads), and a probe
			// takes two slots at once. KMAHIP_TPOS_DENSE: 2/3 as in rounds 1 and 2
			while((1ull << lg) * 2 < (uint64_t) nk * (getenv("KMAHIP_TPOS_DENSE") ? 3 : 6)) ++lg;
			pshift[t] = 32 - lg;
			poff[t + 1] = poff[t] + (1ll << lg);
		}
		poff[1] = poff[0] = 0;
		for(uint32_t t = 1; t < DB_size; ++t) poff[t + 1] = poff[t] + (1ll << (32 - pshift[t]));
		{
			int64_t total = db->h_cat_off[DB_size];
			const size_t n_ps = (size_t) poff[DB_size] + 1;
			unsigned long long *d_ps = nullptr, *d_k = nullptr, *d_ks = nullptr;
			int32_t *d_v = nullptr, *d_vs = nullptr, *d_dups = nullptr;
			int64_t *d_need = nullptr, *d_doff = nullptr, *d_poff = nullptr;
			uint32_t *d_pshift = nullptr;
			void *d_tmp = nullptr;
			std::vector<void *> scratch;
			auto room = [&](void **q, size_t bytes) { if(hipMalloc(q, bytes ? bytes : 16) != hipSuccess) return false; scratch.push_back(*q); return true; };
			struct Free { std::vector<void *> &v; ~Free() { for(void *q : v) (void) hipFree(q); } } guard{scratch};
			if((rc = upload(db, (const unsigned long long *) nullptr, n_ps, (const unsigned long long **) &d_ps)) ||
			   (rc = upload(db, poff.data(), poff.size(), (const int64_t **) &d_poff)) || (rc = upload(db, pshift.data(), pshift.size(), (const uint32_t **) &d_pshift))) { kmahip_db_close(db); return rc; }
			d.tpos_slots = (const uint2 *) d_ps; d.tpos_off = d_poff; d.tpos_shift = d_pshift;
			const size_t T = (size_t) total;
			bool bad = hipMemsetAsync(d_ps, 0, n_ps * 8, 0) != hipSuccess ||
			           !room((void **) &d_k, (T + 1) * 8) || !room((void **) &d_ks, (T + 1) * 8) || !room((void **) &d_v, (T + 1) * 4) || !room((void **) &d_vs, (T + 1) * 4) ||
			           !room((void **) &d_need, (T + 1) * 8) || !room((void **) &d_doff, (T + 1) * 8);
			int64_t n_dups = 0;
			if(!bad && total > 0) {
				hipLaunchKernelGGL(tpos_keys_kernel, dim3(DB_size - 1, grid_y), dim3(256), 0, 0, d.cat, d.cat_off, d.tlen, k, d_k, d_v);
				size_t tmp_bytes = 0, tmp2 = 0;
				bad = rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_k, d_ks, d_v, d_vs, T, 0, 64, 0) != hipSuccess ||
				      rocprim::exclusive_scan(nullptr, tmp2, d_need, d_doff, (int64_t) 0, T + 1, rocprim::plus<int64_t>(), 0) != hipSuccess;
				tmp_bytes = std::max(tmp_bytes, tmp2);
				bad = bad || !room(&d_tmp, tmp_bytes);
				bad = bad || rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_k, d_ks, d_v, d_vs, T, 0, 64, 0) != hipSuccess;
				if(!bad) {
					hipLaunchKernelGGL(tpos_runs_kernel, dim3((unsigned) ((T + 256) / 256)), dim3(256), 0, 0, d_ks, (int64_t) T, d_need);
					bad = rocprim::exclusive_scan(d_tmp, tmp_bytes, d_need, d_doff, (int64_t) 0, T + 1, rocprim::plus<int64_t>(), 0) != hipSuccess ||
					      hipMemcpy(&n_dups, d_doff + T, 8, hipMemcpyDeviceToHost) != hipSuccess;
				}
			}
			if(!bad && n_dups + 1 >= 0x7FFFFFFFll) { kmahip_db_close(db); kmahip_set_error("too many repeated k-mers inside templates"); return KMAHIP_EFORMAT; }
			if(!bad && (rc = upload(db, (const int32_t *) nullptr, (size_t) n_dups + 1, (const int32_t **) &d_dups))) { kmahip_db_close(db); return rc; }
			if(!bad) {
				d.tpos_dups = d_dups;
				bad = hipMemsetAsync(d_dups, 0, 4, 0) != hipSuccess;
				if(!bad && total > 0) hipLaunchKernelGGL(tpos_insert_kernel, dim3((unsigned) ((T + 255) / 256)), dim3(256), 0, 0, d_ks, d_vs, (int64_t) T, d_need, d_doff, d_poff, d_pshift, d_ps, d_dups);
				bad = bad || hipDeviceSynchronize() != hipSuccess;
			}
			if(bad) { kmahip_db_close(db); kmahip_set_error("building the position index failed: %s", hipGetErrorString(hipGetLastError())); return KMAHIP_EDEVICE; }
		}
		std::vector<uint4> tmeta((size_t) 2 * DB_size, make_uint4(0u, 0u, 0u, 0u));
		for(uint32_t t = 1; t < DB_size; ++t) {
			const uint64_t so = (uint64_t) off[t], po = (uint64_t) poff[t];
			tmeta[2 * (size_t) t] = make_uint4((uint32_t) so, (uint32_t) (so >> 32), (uint32_t) po, (uint32_t) (po >> 32));
			tmeta[2 * (size_t) t + 1] = make_uint4((uint32_t) db->h_tlen[t], pshift[t], 0u, 0u);
		}
		if((rc = upload(db, tmeta.data(), tmeta.size(), &d.tmeta))) { kmahip_db_close(db); return rc; }
	}
	stamp("per-template position index");
	*out = db;
	return KMAHIP_OK;
}

extern "C" void kmahip_db_close(kmahip_db *db) {
	if(!db) return;
	for(void *p : db->allocs) (void) hipFree(p);
	delete db;
}

extern "C" int kmahip_db_get_info(const kmahip_db *db, kmahip_db_info *info) {
	if(!db || !info) return KMAHIP_EINVAL;
	*info = db->info;
	return KMAHIP_OK;
}
