// index.hip -- `kma index -i templates.fsa -o prefix [-k k]`: the four files the mapping path reads.
// What the reference does in index.c / makeindex.c:167-330 (makeDB: FileBuffgetFsa, compDNAref, lengthCheck, update_DB,
// updateAnnots) and compress.c:83-614 (compressKMA_DB: bucket directory, key / value-index arrays, value lists de-duplicated by
// valuesHash) restated around a device sort: every k-mer start of every template becomes a 64-bit key (k-mer << 32 | template),
// the keys are radix-sorted and made unique on the GPU, and what is left is one linear pass on the host (group by k-mer, share
// equal template lists, cut into buckets). The files hold the same k-mer -> template-list mapping as the reference's and load into
// it; bucket count, key order inside a bucket and the order of the shared lists are the builder's own (the reference's follow
// from the growth history of its in-memory hash table and carry no meaning for a reader).
// Covered: the default hashed form, k <= 16, no prefix (-Sparse), no minimizers / homopolymer compression, no -batch / -deCon.
#include "kmahip_internal.h"
#include <zlib.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct Tmpl {
	std::string name;
	size_t at = 0;       // first base in the concatenated code array
	int len = 0;
};

// index.c:128-170
struct RefTable {
	uint8_t t[256];
	RefTable() {
		memset(t, 8, sizeof t);
		t[(int) '\n'] = 16;
		const char *codes[5] = {"ARMDrmd", "CYBcyb", "GSKVgskv", "TWHUtwhu", "NXnx"};
		for(int c = 0; c < 5; ++c) for(const char *p = codes[c]; *p; ++p) t[(int) (unsigned char) *p] = (uint8_t) c;
		t[(int) 'a'] = 0;
	}
};

// one thread per base of the concatenated templates: the k-mer that starts there, if it lies inside one template and holds no N
__global__ __launch_bounds__(256) void index_kmers_kernel(const uint8_t *codes, const int64_t *t_off, int n_t, int64_t total, int k,
                                                            unsigned long long *keys) {
	const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(p >= total) return;
	int lo = 0, hi = n_t;                      // template holding p: last t with t_off[t] <= p
	while(hi - lo > 1) { const int mid = (lo + hi) >> 1; if(t_off[mid] <= p) lo = mid; else hi = mid; }
	unsigned long long key = ~0ull;
	if(p + k <= t_off[lo + 1]) {
		unsigned long long km = 0;
		bool ok = true;
		for(int i = 0; i < k; ++i) { const uint8_t c = codes[p + i]; ok = ok && c < 4; km = (km << 2) | (c & 3); }
		if(ok) key = (km << 32) | (unsigned long long) (lo + 1);
	}
	keys[p] = key;
}

struct SameKey { __device__ bool operator()(unsigned long long a, unsigned long long b) const { return a == b; } };

int read_whole(const char *path, std::vector<uint8_t> &buf) {
	gzFile f = gzopen(path, "rb");
	if(!f) { kmahip_set_error("cannot open %s", path); return KMAHIP_EIO; }
	gzbuffer(f, 1 << 20);
	size_t n = buf.size();
	for(;;) {
		buf.resize(n + (8u << 20));
		const int got = gzread(f, buf.data() + n, 8u << 20);
		if(got <= 0) break;
		n += (size_t) got;
	}
	buf.resize(n);
	gzclose(f);
	return KMAHIP_OK;
}

}  // namespace

extern "C" int kmahip_index_build(const char *const *fasta_paths, int n_files, const char *out_prefix, int kmersize) {
	if(!fasta_paths || n_files <= 0 || !out_prefix) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	const int k = kmersize > 0 ? kmersize : 16;
	if(k < 4 || k > 16) { kmahip_set_error("k-mer length %d: 4 ... 16 supported", k); return KMAHIP_EINVAL; }
	static const RefTable T;
	std::vector<Tmpl> tm;
	std::vector<uint8_t> codes;
	for(int fi = 0; fi < n_files; ++fi) {
		std::vector<uint8_t> raw;
		const int rc = read_whole(fasta_paths[fi], raw);
		if(rc) return rc;
		size_t p = 0;
		if(raw.empty() || raw[0] != '>') { kmahip_set_error("%s is not a FASTA file", fasta_paths[fi]); return KMAHIP_EFORMAT; }
		while(p < raw.size()) {
			// header line, chomped (FileBuffgetFsa, seqparse.c:66-159); the sequence is every byte the table knows up to the next '>'
			size_t e = p;
			while(e < raw.size() && raw[e] != '\n') ++e;
			size_t h = e;
			while(h > p + 1 && isspace(raw[h - 1])) --h;
			std::string name((const char *) raw.data() + p + 1, h - p - 1);
			p = e < raw.size() ? e + 1 : e;
			const size_t at = codes.size();
			while(p < raw.size() && raw[p] != '>') { const uint8_t c = T.t[raw[p]]; if(c < 8) codes.push_back(c); ++p; }
			// compDNAref, compdna.c:129-147: leading and trailing N's go; the name says how many led (makeindex.c:229-233)
			size_t a = at, b = codes.size();
			while(a < b && codes[a] == 4) ++a;
			while(b > a && codes[b - 1] == 4) --b;
			const int bias = (int) (a - at);
			if(a > at) memmove(codes.data() + at, codes.data() + a, b - a);
			codes.resize(at + (b - a));
			const int len = (int) (b - a);
			if(len < k) { codes.resize(at); continue; }                  // lengthCheck, qualcheck.c:31-41: "# Skipped"
			if(bias > 0) name += " B" + std::to_string(bias);
			Tmpl t; t.name = name; t.at = at; t.len = len;
			tm.push_back(t);
		}
	}
	const int n_t = (int) tm.size();
	if(n_t == 0) { kmahip_set_error("no template of at least %d bases", k); return KMAHIP_EFORMAT; }
	const uint32_t DB_size = (uint32_t) n_t + 1;
	const int64_t total = (int64_t) codes.size();

	// device: keys, sort, unique
	std::vector<int64_t> t_off((size_t) n_t + 1);
	for(int t = 0; t < n_t; ++t) t_off[(size_t) t] = (int64_t) tm[(size_t) t].at;
	t_off[(size_t) n_t] = total;
	uint8_t *d_codes = nullptr; int64_t *d_off = nullptr;
	unsigned long long *d_keys = nullptr, *d_sorted = nullptr, *d_uniq = nullptr; size_t *d_count = nullptr; void *d_tmp = nullptr;
	auto release = [&] { (void) hipFree(d_codes); (void) hipFree(d_off); (void) hipFree(d_keys); (void) hipFree(d_sorted); (void) hipFree(d_uniq); (void) hipFree(d_count); (void) hipFree(d_tmp); };
	struct Guard { decltype(release) &r; ~Guard() { r(); } } guard{release};
	HIP_TRY(hipMalloc((void **) &d_codes, (size_t) total + 32));
	HIP_TRY(hipMalloc((void **) &d_off, t_off.size() * 8));
	HIP_TRY(hipMalloc((void **) &d_keys, (size_t) total * 8));
	HIP_TRY(hipMalloc((void **) &d_sorted, (size_t) total * 8));
	HIP_TRY(hipMalloc((void **) &d_uniq, (size_t) total * 8));
	HIP_TRY(hipMalloc((void **) &d_count, sizeof(size_t)));
	HIP_TRY(hipMemcpy(d_codes, codes.data(), (size_t) total, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_off, t_off.data(), t_off.size() * 8, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(index_kmers_kernel, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, 0, d_codes, d_off, n_t, total, k, d_keys);
	HIP_TRY(hipGetLastError());
	size_t tmp_bytes = 0, tmp2 = 0;
	if(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_keys, d_sorted, (size_t) total, 0, 64, 0) != hipSuccess ||
	   rocprim::unique(nullptr, tmp2, d_sorted, d_uniq, d_count, (size_t) total, SameKey(), 0) != hipSuccess) { kmahip_set_error("rocprim size query failed"); return KMAHIP_EDEVICE; }
	tmp_bytes = std::max(tmp_bytes, tmp2);
	HIP_TRY(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
	if(rocprim::radix_sort_keys(d_tmp, tmp_bytes, d_keys, d_sorted, (size_t) total, 0, 64, 0) != hipSuccess ||
	   rocprim::unique(d_tmp, tmp_bytes, d_sorted, d_uniq, d_count, (size_t) total, SameKey(), 0) != hipSuccess) { kmahip_set_error("rocprim sort / unique failed"); return KMAHIP_EDEVICE; }
	HIP_TRY(hipDeviceSynchronize());
	size_t n_pairs = 0;
	HIP_TRY(hipMemcpy(&n_pairs, d_count, sizeof n_pairs, hipMemcpyDeviceToHost));
	std::vector<unsigned long long> pairs(n_pairs);
	if(n_pairs) HIP_TRY(hipMemcpy(pairs.data(), d_uniq, n_pairs * 8, hipMemcpyDeviceToHost));
	if(n_pairs && pairs.back() == ~0ull) pairs.pop_back();         // the starts that hold no k-mer sorted to the end
	if(pairs.empty()) { kmahip_set_error("the templates hold no k-mer"); return KMAHIP_EFORMAT; }

	// host: group by k-mer, share equal lists (valuesHash, compress.c:218), lists laid out in the order they are first needed
	const bool u16 = DB_size < 65535;                               // hashmapkma.c:340-348
	std::vector<uint32_t> ukm, vi_of_key;
	std::vector<uint32_t> values;                                   // [cnt, t1 .. tcnt] per list
	std::unordered_map<uint64_t, std::vector<uint32_t>> by_sig;     // signature -> offsets of lists with it
	std::vector<uint32_t> list;
	for(size_t a = 0; a < pairs.size();) {
		size_t b = a;
		const uint32_t km = (uint32_t) (pairs[a] >> 32);
		list.clear();
		uint64_t sig = 0x9E3779B97F4A7C15ull;
		for(; b < pairs.size() && (uint32_t) (pairs[b] >> 32) == km; ++b) {
			const uint32_t t = (uint32_t) pairs[b];
			list.push_back(t);
			sig = (sig ^ t) * 0xBF58476D1CE4E5B9ull; sig ^= sig >> 29;
		}
		uint32_t off = 0xFFFFFFFFu;
		std::vector<uint32_t> &cand = by_sig[sig];
		for(uint32_t o : cand) {
			if(values[o] == list.size() && !memcmp(&values[o + 1], list.data(), list.size() * 4)) { off = o; break; }
		}
		if(off == 0xFFFFFFFFu) {
			if(values.size() + list.size() + 1 >= 0xFFFFFFFFull) { kmahip_set_error("value lists exceed 32-bit offsets"); return KMAHIP_EFORMAT; }
			off = (uint32_t) values.size();
			values.push_back((uint32_t) list.size());
			values.insert(values.end(), list.begin(), list.end());
			cand.push_back(off);
		}
		ukm.push_back(km); vi_of_key.push_back(off);
		a = b;
	}
	const uint64_t n = ukm.size(), v_index = values.size();
	// buckets: kpos = key & (size - 1) (hashMap_getGlobal, hashmapkma.c:149-178); never the direct-address form
	uint64_t size = 1u << 20;
	while(size < n) size <<= 1;
	const uint64_t kspace = 1ull << (2 * k);
	if(size >= kspace) size = kspace >> 1;
	if(size < 2 || n > 0xFFFFFFFFull) { kmahip_set_error("index shape not supported"); return KMAHIP_EFORMAT; }
	std::vector<uint32_t> exist(size, (uint32_t) n), fill(size + 1, 0);
	for(uint64_t i = 0; i < n; ++i) ++fill[(ukm[i] & (size - 1)) + 1];
	for(uint64_t b = 0; b < size; ++b) { if(fill[b + 1]) exist[b] = fill[b]; fill[b + 1] += fill[b]; }
	std::vector<uint32_t> key_index(n + 1), value_index(n);
	{
		std::vector<uint32_t> at(fill.begin(), fill.end() - 1);
		for(uint64_t i = 0; i < n; ++i) { const uint32_t s = at[ukm[i] & (size - 1)]++; key_index[s] = ukm[i]; value_index[s] = vi_of_key[i]; }
	}
	// the sentinel behind the last key ends the scan of the last bucket: any key of another bucket (compress.c:549-585)
	key_index[n] = (uint32_t) (((key_index[n - 1] & (size - 1)) + 1) & (size - 1));

	const std::string base(out_prefix);
	FILE *f = fopen((base + ".comp.b").c_str(), "wb");
	if(!f) { kmahip_set_error("cannot create %s.comp.b", out_prefix); return KMAHIP_EIO; }
	const uint32_t h32[3] = {DB_size, (uint32_t) k, 0};
	const uint64_t h64[5] = {0, size, n, v_index, n};
	bool ok = fwrite(h32, 4, 3, f) == 3 && fwrite(h64, 8, 5, f) == 5 && fwrite(exist.data(), 4, size, f) == size;
	if(u16) {
		std::vector<uint16_t> v16(values.begin(), values.end());
		ok = ok && fwrite(v16.data(), 2, v16.size(), f) == v16.size();
	} else ok = ok && fwrite(values.data(), 4, values.size(), f) == values.size();
	const uint32_t tail[2] = {(uint32_t) k, 0};
	ok = ok && fwrite(key_index.data(), 4, n + 1, f) == n + 1 && fwrite(value_index.data(), 4, n, f) == n && fwrite(tail, 4, 2, f) == 2;
	ok = (fclose(f) == 0) && ok;
	// .length.b: DB_size, then the lengths with slot 0 = the k of the position index (makeindex.c:300-311)
	f = fopen((base + ".length.b").c_str(), "wb");
	if(!f) { kmahip_set_error("cannot create %s.length.b", out_prefix); return KMAHIP_EIO; }
	std::vector<uint32_t> lens(DB_size);
	lens[0] = (uint32_t) k;
	for(int t = 0; t < n_t; ++t) lens[(size_t) t + 1] = (uint32_t) tm[(size_t) t].len;
	ok = fwrite(&DB_size, 4, 1, f) == 1 && fwrite(lens.data(), 4, DB_size, f) == DB_size && ok;
	ok = (fclose(f) == 0) && ok;
	// .seq.b: (len >> 5) + 1 words per template, 32 bases a word from the top, N as A (updateAnnots; read back at runkma.c:214-220)
	f = fopen((base + ".seq.b").c_str(), "wb");
	if(!f) { kmahip_set_error("cannot create %s.seq.b", out_prefix); return KMAHIP_EIO; }
	std::vector<uint64_t> words;
	for(int t = 0; t < n_t; ++t) {
		const Tmpl &x = tm[(size_t) t];
		words.assign((size_t) (x.len >> 5) + 1, 0);
		for(int i = 0; i < x.len; ++i) words[(size_t) i >> 5] |= (uint64_t) (codes[x.at + (size_t) i] & 3) << (62 - ((i & 31) << 1));
		ok = fwrite(words.data(), 8, words.size(), f) == words.size() && ok;
	}
	ok = (fclose(f) == 0) && ok;
	f = fopen((base + ".name").c_str(), "wb");
	if(!f) { kmahip_set_error("cannot create %s.name", out_prefix); return KMAHIP_EIO; }
	for(int t = 0; t < n_t; ++t) ok = fprintf(f, "%s\n", tm[(size_t) t].name.c_str()) > 0 && ok;
	ok = (fclose(f) == 0) && ok;
	if(!ok) { kmahip_set_error("writing the index under %s failed", out_prefix); return KMAHIP_EIO; }
	return KMAHIP_OK;
}
