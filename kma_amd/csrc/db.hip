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
	p->scoreT = 0.5; p->mrc = 0.0; p->minFrac = 1.0;
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

// one workgroup per template; slots[].y still holds the list offsets
__global__ __launch_bounds__(256) void walk_vsid_kernel(const uint2 *slots, uint32_t nb_log2, const uint64_t *cat, const int64_t *cat_off, const int32_t *tlen,
                                                        int k, uint32_t *vs_id, uint32_t *first) {
	const uint32_t t = blockIdx.x + 1;
	const int64_t g0 = cat_off[t];
	const int tl = tlen[t];
	for(int i = threadIdx.x; i + k <= tl; i += blockDim.x) {
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
	std::vector<uint2> slots(nb * KMAHIP_BUCKET_SLOTS, make_uint2(0u, KMAHIP_EMPTY_VI));
	for(uint64_t i = 0; i < n; ++i) {
		uint64_t b = home_bucket(keys[i], nb_log2);
		for(;;) {
			uint2 *s = &slots[b * KMAHIP_BUCKET_SLOTS];
			int j = 0;
			while(j < KMAHIP_BUCKET_SLOTS && s[j].y != KMAHIP_EMPTY_VI) ++j;
			if(j < KMAHIP_BUCKET_SLOTS) { s[j] = make_uint2(keys[i], vidx[i]); break; }
			b = (b + 1) & (nb - 1);
		}
	}
	db->info.hash_bytes = slots.size() * sizeof(uint2);
	// presence bits: ~7 bits per k-mer (13 % false positives) as long as that fits 2 MiB
	std::vector<uint32_t> kbits;
	uint32_t kbits_log2 = 0;
	if(n > 0 && 8 * n < (1ull << 25) && !getenv("KMAHIP_NO_KBITS")) {        // (the switch exists for the test that compares both paths)
		kbits_log2 = 16;
		while((2ull << kbits_log2) <= 8 * n) ++kbits_log2;
		kbits.assign((size_t) 1 << (kbits_log2 - 5), 0u);
		for(uint64_t i = 0; i < n; ++i) {
			const uint32_t h = (keys[i] * KMAHIP_KBITS_MUL) >> (32 - kbits_log2);
			kbits[h >> 5] |= 1u << (h & 31);
		}
	}

	int rc;
	DevDB &d = db->dev;
	memset(&d, 0, sizeof d);
	d.DB_size = DB_size; d.kmersize = tail[0]; d.mlen = mlen; d.nb_log2 = nb_log2; d.values_u16 = u16;
	// (slots are uploaded further down, once every key has been given a template position)
	if(!kbits.empty()) {
		if((rc = upload(db, kbits.data(), kbits.size(), &d.kbits))) { kmahip_db_close(db); return rc; }
		d.kbits_shift = 32 - kbits_log2;
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
			uint2 *d_slots = nullptr;
			uint32_t *d_vsid = nullptr, *d_first = nullptr;
			unsigned long long *d_unplaced = nullptr, unplaced = 0;
			if((rc = upload(db, slots.data(), slots.size(), (const uint2 **) &d_slots)) || (rc = upload(db, cat.data(), cat.size(), &d.cat)) ||
			   (rc = upload(db, db->h_cat_off.data(), db->h_cat_off.size(), &d.cat_off)) ||
			   (rc = upload(db, (const uint32_t *) nullptr, (size_t) total + 64, (const uint32_t **) &d_vsid))) { kmahip_db_close(db); return rc; }
			d.slots = d_slots; d.vs_id = d_vsid;
			bool bad = hipMalloc((void **) &d_first, slots.size() * 4 + 8) != hipSuccess;
			if(!bad) {
				d_unplaced = (unsigned long long *) (d_first + slots.size());
				bad = hipMemsetAsync(d_first, 0xFF, slots.size() * 4, 0) != hipSuccess || hipMemsetAsync(d_unplaced, 0, 8, 0) != hipSuccess ||
				      hipMemsetAsync(d_vsid, 0xFF, ((size_t) total + 64) * 4, 0) != hipSuccess;
				if(!bad && DB_size > 1) {
					hipLaunchKernelGGL(walk_vsid_kernel, dim3(DB_size - 1), dim3(256), 0, 0, d_slots, nb_log2, d.cat, d.cat_off, d.tlen, kk, d_vsid, d_first);
					hipLaunchKernelGGL(walk_repoint_kernel, dim3((unsigned) ((slots.size() + 255) / 256)), dim3(256), 0, 0, d_slots, (int64_t) slots.size(), d_first, d_unplaced);
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
			while((1ull << lg) * 2 < (uint64_t) nk * 3) ++lg;   // load <= 2/3
			pshift[t] = 32 - lg;
			poff[t + 1] = poff[t] + (1ll << lg);
		}
		poff[1] = poff[0] = 0;
		for(uint32_t t = 1; t < DB_size; ++t) poff[t + 1] = poff[t] + (1ll << (32 - pshift[t]));
		std::vector<uint2> pslots((size_t) poff[DB_size] + 1, make_uint2(0u, 0u));
		// built by a few threads, each over a stretch of templates with a duplicate list of its own; the lists are then put
		// one after the other and the references into them moved by where each list landed
		std::vector<int32_t> dups(1, 0);
		{
			const char *e = getenv("KMAHIP_IO_THREADS");
			const int hw = (int) std::thread::hardware_concurrency();
			int nt = e ? atoi(e) : std::min(16, hw > 0 ? hw : 1);
			nt = std::max(1, std::min<int>(nt, (int) (DB_size / 64) + 1));
			std::vector<std::vector<int32_t>> part((size_t) nt);
			std::vector<uint32_t> cut((size_t) nt + 1, DB_size);
			{	// equal shares of the template bases
				int64_t total = 0, acc = 0;
				for(uint32_t t = 1; t < DB_size; ++t) total += db->h_tlen[t];
				cut[0] = 1;
				int c = 1;
				for(uint32_t t = 1; t < DB_size && c < nt; ++t) {
					acc += db->h_tlen[t];
					if(acc * nt >= total * c) cut[(size_t) c++] = t + 1;
				}
			}
			auto build = [&](int w) {
				std::vector<std::pair<uint32_t, int32_t>> kp;
				std::vector<int32_t> &dl = part[(size_t) w];
				for(uint32_t t = cut[(size_t) w]; t < cut[(size_t) w + 1]; ++t) {
					const int tl = db->h_tlen[t];
					const uint64_t *ts = tseq.data() + off[t];
					kp.clear();
					for(int i = 0; i + k <= tl; ++i) {
						const int ip = (i & 31) << 1, wd = i >> 5;
						uint64_t x = ts[wd] << ip;
						if(ip) x |= ts[wd + 1] >> (64 - ip);
						const uint32_t km = (uint32_t) (x >> (64 - 2 * k));
						if(km) kp.push_back({km, i + 1});
					}
					std::sort(kp.begin(), kp.end());
					const uint32_t sh = pshift[t];
					const uint64_t msk = (1ull << (32 - sh)) - 1;
					uint2 *tab = pslots.data() + poff[t];
					for(size_t a = 0; a < kp.size();) {
						size_t b = a;
						while(b < kp.size() && kp[b].first == kp[a].first) ++b;
						int32_t val;
						if(b - a == 1) val = kp[a].second;
						else {
							val = -((int32_t) dl.size() + 1);         // (relative to this thread's list for now)
							dl.push_back((int32_t) (b - a));
							for(size_t c = a; c < b; ++c) dl.push_back(kp[c].second);
						}
						uint64_t sl = (uint32_t) (kp[a].first * 0x9E3779B1u) >> sh;
						while(tab[sl].y != 0) sl = (sl + 1) & msk;
						tab[sl] = make_uint2(kp[a].first, (uint32_t) val);
						a = b;
					}
				}
			};
			{
				std::vector<std::thread> pool;
				for(int w = 1; w < nt; ++w) pool.emplace_back(build, w);
				build(0);
				for(std::thread &th : pool) th.join();
			}
			std::vector<int64_t> base((size_t) nt + 1, 1);
			for(int w = 0; w < nt; ++w) base[(size_t) w + 1] = base[(size_t) w] + (int64_t) part[(size_t) w].size();
			if(base[(size_t) nt] >= 0x7FFFFFFFll) { kmahip_db_close(db); kmahip_set_error("too many repeated k-mers inside templates"); return KMAHIP_EFORMAT; }
			dups.resize((size_t) base[(size_t) nt]);
			auto place = [&](int w) {
				if(!part[(size_t) w].empty()) memcpy(dups.data() + base[(size_t) w], part[(size_t) w].data(), part[(size_t) w].size() * sizeof(int32_t));
				const int32_t shift = (int32_t) base[(size_t) w];
				if(!shift) return;
				for(int64_t i = poff[cut[(size_t) w]]; i < poff[cut[(size_t) w + 1]]; ++i) {
					const int32_t v = (int32_t) pslots[(size_t) i].y;
					if(v < 0) pslots[(size_t) i].y = (uint32_t) (v - shift);
				}
			};
			{
				std::vector<std::thread> pool;
				for(int w = 1; w < nt; ++w) pool.emplace_back(place, w);
				place(0);
				for(std::thread &th : pool) th.join();
			}
		}
		std::vector<uint4> tmeta((size_t) 2 * DB_size, make_uint4(0u, 0u, 0u, 0u));
		for(uint32_t t = 1; t < DB_size; ++t) {
			const uint64_t so = (uint64_t) off[t], po = (uint64_t) poff[t];
			tmeta[2 * (size_t) t] = make_uint4((uint32_t) so, (uint32_t) (so >> 32), (uint32_t) po, (uint32_t) (po >> 32));
			tmeta[2 * (size_t) t + 1] = make_uint4((uint32_t) db->h_tlen[t], pshift[t], 0u, 0u);
		}
		if((rc = upload(db, tmeta.data(), tmeta.size(), &d.tmeta)) ||
		   (rc = upload(db, pslots.data(), pslots.size(), &d.tpos_slots)) ||
		   (rc = upload(db, poff.data(), poff.size(), &d.tpos_off)) ||
		   (rc = upload(db, pshift.data(), pshift.size(), &d.tpos_shift)) ||
		   (rc = upload(db, dups.data(), dups.size(), &d.tpos_dups))) { kmahip_db_close(db); return rc; }
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
