// fastgz.h -- gzip members written with Huffman coding only (RFC 1951 dynamic blocks without matches, RFC 1952 wrapper), host code.
// The reference compresses its outputs with zlib at level 1 (deflateInit2 in filebuff.c:189); what a reader gets back after
// inflating is the contract, not the compressed bytes. The rows of a .frag.gz are DNA text with few repeats longer than the
// 3-byte minimum a match needs, so an encoder that only entropy-codes the bytes gets within ~20 % of zlib's level-1 size at
// several times its speed, and it runs on as many threads as there are blocks of rows.
#pragma once
#include <zlib.h>       // crc32()
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
	code_lengths(freq, 257, 15, llen);
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
	// the literals: code and length of a byte in one table entry, two bytes per turn of the loop
	uint32_t tab[256];
	for(int s = 0; s < 256; ++s) tab[s] = (uint32_t) lcode[s] | ((uint32_t) llen[s] << 16);
	size_t i = 0;
	for(; i + 2 <= n; i += 2) {
		const uint32_t a = tab[p[i]], b = tab[p[i + 1]];
		const int la = (int) (a >> 16), lb = (int) (b >> 16);
		bs.put((a & 0xffff) | ((b & 0xffff) << la), la + lb);
	}
	if(i < n) bs.put(tab[p[i]] & 0xffff, (int) (tab[p[i]] >> 16));
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
	uint32_t crc = (uint32_t) crc32(0L, Z_NULL, 0);
	for(size_t a = 0; a < n; a += 1u << 30) crc = (uint32_t) crc32(crc, p + a, (uInt) std::min<size_t>(1u << 30, n - a));
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
