// fastgz.h -- gzip members written with Huffman coding only (RFC 1951 dynamic blocks without matches, RFC 1952 wrapper), host code.
// The reference compresses its outputs with zlib at level 1 (deflateInit2 in filebuff.c:189); what a reader gets back after
// inflating is the contract, not the compressed bytes. The rows of a .frag.gz are DNA text with few repeats longer than the
// 3-byte minimum a match needs, so an encoder that only entropy-codes the bytes gets within ~20 % of zlib's level-1 size at
// several times its speed, and it runs on as many threads as there are blocks of rows.
#pragma once
#include <zlib.h>       // crc32()
#include <immintrin.h>  // the carry-less multiply CRC below (x86-64 hosts; zlib's table form is the fallback)
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace fastgz {

// code lengths of a Huffman code over n symbols with counts freq[], none longer than `limit` bits. Symbols with count 0 get
// length 0; at least two symbols get a code (RFC 1951 readers want a complete code).
inline void code_lengths(const uint32_t *freq_in, int n, int limit, uint8_t *len) {
	std::vector<uint64_t> freq(freq_in, freq_in + n);
	int used = 0;
	for(int i = 0; i < n; ++i) used += freq[(size_t) i] != 0;
	for(int i = 0; used < 2 && i < n; ++i) if(!freq[(size_t) i]) { freq[(size_t) i] = 1; ++used; }
	std::vector<int> order;
	for(;;) {
		// the two-queue construction over the symbols sorted by count
		order.clear();
		for(int i = 0; i < n; ++i) if(freq[(size_t) i]) order.push_back(i);
		std::sort(order.begin(), order.end(), [&](int a, int b) { return freq[(size_t) a] != freq[(size_t) b] ? freq[(size_t) a] < freq[(size_t) b] : a < b; });
		const int m = (int) order.size();
		std::vector<uint64_t> w((size_t) (2 * m));
		std::vector<int> parent((size_t) (2 * m), -1);
		for(int i = 0; i < m; ++i) w[(size_t) i] = freq[(size_t) order[(size_t) i]];
		int leaf = 0, node = m, made = m;
		auto take = [&]() {
			if(leaf < m && (node >= made || w[(size_t) leaf] <= w[(size_t) node])) return leaf++;
			return node++;
		};
		while(made < 2 * m - 1) {
			const int a = take(), b = take();
			w[(size_t) made] = w[(size_t) a] + w[(size_t) b];
			parent[(size_t) a] = parent[(size_t) b] = made;
			++made;
		}
		int deepest = 0;
		std::vector<int> depth((size_t) (2 * m), 0);
		for(int i = 2 * m - 3; i >= 0; --i) depth[(size_t) i] = depth[(size_t) parent[(size_t) i]] + 1;
		memset(len, 0, (size_t) n);
		for(int i = 0; i < m; ++i) { len[order[(size_t) i]] = (uint8_t) depth[(size_t) i]; deepest = std::max(deepest, depth[(size_t) i]); }
		if(deepest <= limit) return;
		for(int i = 0; i < n; ++i) if(freq[(size_t) i]) freq[(size_t) i] = (freq[(size_t) i] + 1) >> 1;     // flatter counts, shallower tree
	}
}

// canonical codes (RFC 1951 3.2.2), bit-reversed: deflate packs Huffman codes starting from their most significant bit
inline void canonical_codes(const uint8_t *len, int n, uint16_t *code) {
	int count[16] = {0}, next[16] = {0};
	for(int i = 0; i < n; ++i) ++count[len[i]];
	count[0] = 0;
	for(int b = 1, c = 0; b < 16; ++b) { c = (c + count[b - 1]) << 1; next[b] = c; }
	for(int i = 0; i < n; ++i) {
		const int l = len[i];
		if(!l) { code[i] = 0; continue; }
		unsigned c = (unsigned) next[l]++, r = 0;
		for(int b = 0; b < l; ++b) { r = (r << 1) | (c & 1); c >>= 1; }
		code[i] = (uint16_t) r;
	}
}

// CRC-32 (the gzip polynomial) by folding with carry-less multiplication: 64 bytes per turn of the loop, an order of magnitude
// faster than the table form zlib 1.2 ships -- the checksum was a third of a member's cost. The constants are the usual ones for
// this polynomial in bit-reflected form (x^(n) mod P for the fold distances; "Fast CRC Computation for Generic Polynomials Using
// PCLMULQDQ", Gopal et al.). tests/test_fastgz.py compares it with zlib's crc32 on every length from 0 to 300 and on megabytes.
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1")))
inline uint32_t crc32_fold(uint32_t crc, const uint8_t *buf, size_t len) {
	// (crc: the running value as zlib's crc32() takes and returns it)
	if(len < 64) return (uint32_t) ::crc32(crc, buf, (uInt) len);
	const size_t body = len & ~(size_t) 15;
	alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
	alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
	alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
	alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
	__m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
	size_t left = body;
	x1 = _mm_loadu_si128((const __m128i *) (buf + 0x00));
	x2 = _mm_loadu_si128((const __m128i *) (buf + 0x10));
	x3 = _mm_loadu_si128((const __m128i *) (buf + 0x20));
	x4 = _mm_loadu_si128((const __m128i *) (buf + 0x30));
	x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int) ~crc));
	x0 = _mm_load_si128((const __m128i *) k1k2);
	buf += 64; left -= 64;
	while(left >= 64) {
		x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
		x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
		x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
		x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
		y5 = _mm_loadu_si128((const __m128i *) (buf + 0x00)); y6 = _mm_loadu_si128((const __m128i *) (buf + 0x10));
		y7 = _mm_loadu_si128((const __m128i *) (buf + 0x20)); y8 = _mm_loadu_si128((const __m128i *) (buf + 0x30));
		x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
		x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
		buf += 64; left -= 64;
	}
	// four accumulators into one
	x0 = _mm_load_si128((const __m128i *) k3k4);
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
	x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
	while(left >= 16) {
		x2 = _mm_loadu_si128((const __m128i *) buf);
		x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
		buf += 16; left -= 16;
	}
	// 128 -> 64 -> 32 bits, Barrett reduction
	x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
	x3 = _mm_setr_epi32(~0, 0, ~0, 0);
	x1 = _mm_srli_si128(x1, 8);
	x1 = _mm_xor_si128(x1, x2);
	x0 = _mm_loadl_epi64((const __m128i *) k5k0);
	x2 = _mm_srli_si128(x1, 4);
	x1 = _mm_and_si128(x1, x3);
	x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
	x1 = _mm_xor_si128(x1, x2);
	x0 = _mm_load_si128((const __m128i *) poly);
	x2 = _mm_and_si128(x1, x3);
	x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
	x2 = _mm_and_si128(x2, x3);
	x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
	x1 = _mm_xor_si128(x1, x2);
	uint32_t c = ~(uint32_t) _mm_extract_epi32(x1, 1);
	if(len > body) c = (uint32_t) ::crc32(c, buf, (uInt) (len - body));          // (buf stands behind the folded part)
	return c;
}
inline bool have_clmul() { static const bool ok = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1"); return ok; }
#else
inline uint32_t crc32_fold(uint32_t crc, const uint8_t *buf, size_t len) { return (uint32_t) ::crc32(crc, buf, (uInt) len); }
inline bool have_clmul() { return false; }
#endif
inline uint32_t crc32_fast(uint32_t crc, const uint8_t *buf, size_t len) {
	if(!have_clmul()) { for(size_t a = 0; a < len; a += 1u << 30) crc = (uint32_t) ::crc32(crc, buf + a, (uInt) std::min<size_t>(1u << 30, len - a)); return crc; }
	return crc32_fold(crc, buf, len);
}

// bits go straight into a buffer the caller has sized for the worst case (15 bits per literal + the block headers): eight
// bytes are stored at every put and the write position moves on by the whole bytes among them
struct BitSink {
	uint8_t *w;
	uint64_t acc = 0;
	int n = 0;
	explicit BitSink(uint8_t *dst) : w(dst) {}
	inline void put(uint32_t v, int bits) {           // bits <= 32, n < 8 on entry
		acc |= (uint64_t) v << n;
		n += bits;
		memcpy(w, &acc, 8);                            // (little endian host)
		w += n >> 3;
		acc >>= n & ~7;
		n &= 7;
	}
	inline void put64(uint64_t v, int bits) {         // bits <= 56, n < 8 on entry
		acc |= v << n;
		n += bits;
		memcpy(w, &acc, 8);
		w += n >> 3;
		acc >>= n & ~7;
		n &= 7;
	}
	uint8_t *finish() { if(n > 0) { *w++ = (uint8_t) acc; } acc = 0; n = 0; return w; }
};

// one deflate block of literals with a Huffman code of its own
inline void deflate_block(BitSink &bs, const uint8_t *p, size_t n, bool last) {
	uint32_t freq[257] = {0};
	{
		uint32_t f4[4][256];
		memset(f4, 0, sizeof f4);
		size_t i = 0;
		for(; i + 4 <= n; i += 4) { ++f4[0][p[i]]; ++f4[1][p[i + 1]]; ++f4[2][p[i + 2]]; ++f4[3][p[i + 3]]; }
		for(; i < n; ++i) ++f4[0][p[i]];
		for(int s = 0; s < 256; ++s) freq[s] = f4[0][s] + f4[1][s] + f4[2][s] + f4[3][s];
	}
	freq[256] = 1;                                   // end of block
	uint8_t llen[257];
	uint16_t lcode[257];
	code_lengths(freq, 257, 12, llen);               // (12 bits at most: four codes fit one put of the bit sink, below)
	canonical_codes(llen, 257, lcode);
	// the two code-length sequences (257 literal/length codes, then two distance codes of one bit that are never used: readers
	// insist on a distance code, deflate.c of zlib sends the same), run-length coded with the code-length alphabet
	uint8_t seq[259];
	memcpy(seq, llen, 257);
	seq[257] = seq[258] = 1;
	struct Tok { uint8_t sym, extra, extra_bits; };
	Tok tok[259];
	int n_tok = 0;
	uint32_t cfreq[19] = {0};
	for(int i = 0; i < 259;) {
		int j = i;
		while(j < 259 && seq[j] == seq[i]) ++j;
		int run = j - i;
		if(seq[i] == 0) {
			while(run >= 11) { const int r = std::min(run, 138); tok[n_tok++] = Tok{18, (uint8_t) (r - 11), 7}; run -= r; }
			if(run >= 3) { tok[n_tok++] = Tok{17, (uint8_t) (run - 3), 3}; run = 0; }
		} else if(run >= 4) {
			tok[n_tok++] = Tok{seq[i], 0, 0};
			--run;
			while(run >= 3) { const int r = std::min(run, 6); tok[n_tok++] = Tok{16, (uint8_t) (r - 3), 2}; run -= r; }
		}
		while(run-- > 0) tok[n_tok++] = Tok{seq[i], 0, 0};
		i = j;
	}
	for(int t = 0; t < n_tok; ++t) ++cfreq[tok[t].sym];
	uint8_t clen[19];
	uint16_t ccode[19];
	code_lengths(cfreq, 19, 7, clen);
	canonical_codes(clen, 19, ccode);
	static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
	int hclen = 19;
	while(hclen > 4 && clen[order[hclen - 1]] == 0) --hclen;
	bs.put(last ? 1u : 0u, 1);
	bs.put(2, 2);                                    // dynamic Huffman
	bs.put(257 - 257, 5);                            // HLIT
	bs.put(2 - 1, 5);                                // HDIST
	bs.put((uint32_t) (hclen - 4), 4);
	for(int i = 0; i < hclen; ++i) bs.put(clen[order[i]], 3);
	for(int t = 0; t < n_tok; ++t) {
		bs.put(ccode[tok[t].sym], clen[tok[t].sym]);
		if(tok[t].extra_bits) bs.put(tok[t].extra, tok[t].extra_bits);
	}
	// the literals: code and length of a byte in one table entry, four bytes per turn of the loop (4 x 12 bits + the 7 that may
	// be waiting in the sink fit its 64)
	uint32_t tab[256];
	for(int s = 0; s < 256; ++s) tab[s] = (uint32_t) lcode[s] | ((uint32_t) llen[s] << 16);
	size_t i = 0;
	for(; i + 4 <= n; i += 4) {
		const uint32_t a = tab[p[i]], b = tab[p[i + 1]], c = tab[p[i + 2]], d = tab[p[i + 3]];
		const int la = (int) (a >> 16), lb = (int) (b >> 16), lc = (int) (c >> 16), ld = (int) (d >> 16);
		bs.put64((uint64_t) (a & 0xffff) | ((uint64_t) (b & 0xffff) << la) | ((uint64_t) (c & 0xffff) << (la + lb)) | ((uint64_t) (d & 0xffff) << (la + lb + lc)), la + lb + lc + ld);
	}
	for(; i < n; ++i) bs.put(tab[p[i]] & 0xffff, (int) (tab[p[i]] >> 16));
	bs.put(lcode[256], llen[256]);
}

// [p, p + n) as one gzip member appended to out (scratch: kept by the caller between calls, any size)
inline void gzip_member(const uint8_t *p, size_t n, std::string &out, std::vector<uint8_t> &scratch) {
	static const unsigned char head[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 4 /* fastest */, 3 /* unix */};
	const size_t BLOCK = 1u << 18;
	const size_t worst = 2 * n + 512 * (n / BLOCK + 1) + 64;
	if(scratch.size() < worst) scratch.resize(worst);
	BitSink bs(scratch.data());
	if(n == 0) deflate_block(bs, p, 0, true);
	for(size_t a = 0; a < n; a += BLOCK) deflate_block(bs, p + a, std::min(BLOCK, n - a), a + BLOCK >= n);
	const size_t bytes = (size_t) (bs.finish() - scratch.data());
	const uint32_t crc = crc32_fast((uint32_t) crc32(0L, Z_NULL, 0), p, n);
	const uint32_t tail[2] = {crc, (uint32_t) (n & 0xffffffffu)};
	out.reserve(out.size() + bytes + 18);
	out.append((const char *) head, 10);
	out.append((const char *) scratch.data(), bytes);
	out.append((const char *) tail, 8);
}

inline void gzip_member(const uint8_t *p, size_t n, std::string &out) {
	std::vector<uint8_t> scratch;
	gzip_member(p, n, out, scratch);
}

}  // namespace fastgz
