/* oracle/db.c -- TEST INFRASTRUCTURE (see kma_oracle.h).
 * Reader for the reference's on-disk index:
 *   <prefix>.comp.b   hashmapkma.c:275-455 (load), :722-775 (dump)
 *   <prefix>.length.b runkma.c:76-93, makeindex.c:263-273
 *   <prefix>.seq.b    updateindex.c:169-182, offsets runkma.c:214-220
 * and the chained-bucket lookup hashMap_getGlobal (hashmapkma.c:149-178).
 */
#include "kma_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void orc_default_rewards(orc_rewards *r) {
	/* kma.c:327-336 defaults, MM recomputed at kma.c:1308 as (Ts+Tv-1)/2,
	 * substitution matrix kma.c:1309-1328 */
	const int Ts = -2, Tv = -2;
	r->M = 1; r->U = -1; r->W1 = -3; r->Wl = -6; r->Mn = 0; r->PE = 7;
	r->MM = (Ts + Tv - 1) / 2;
	for(int i = 0; i < 4; ++i) {
		for(int j = 0; j < 4; ++j) r->d[i][j] = Tv;
		r->d[i][4] = r->Mn;
		r->d[i][i ^ 2] = Ts;   /* A<->G, C<->T are transitions */
		r->d[i][i] = r->M;
	}
	for(int j = 0; j < 5; ++j) r->d[4][j] = r->Mn;
	r->d[4][4] = 0;
}

static void *slurp(FILE *f, size_t bytes) {
	void *p = malloc(bytes ? bytes : 1);
	if(!p) return 0;
	if(fread(p, 1, bytes, f) != bytes) { free(p); return 0; }
	return p;
}

orc_db *orc_db_load(const char *prefix) {
	char path[4096];
	orc_db *db = calloc(1, sizeof(orc_db));
	FILE *f;
	if(!db) return 0;

	snprintf(path, sizeof path, "%s.comp.b", prefix);
	if(!(f = fopen(path, "rb"))) { free(db); return 0; }
	uint32_t h32[3];
	uint64_t h64[5];
	if(fread(h32, 4, 3, f) != 3 || fread(h64, 8, 5, f) != 5) goto fail;
	db->DB_size = h32[0]; db->mlen = h32[1]; db->prefix_len = h32[2];
	db->prefix = h64[0]; db->size = h64[1]; db->n = h64[2];
	db->v_index = h64[3]; db->null_index = h64[4];
	uint64_t kmask = (db->mlen >= 32) ? ~0ull : ((1ull << (2 * db->mlen)) - 1);
	if(db->size - 1 == kmask) goto fail;           /* direct-address "megamap": not restated */
	if(db->n > 0xFFFFFFFFull) goto fail;           /* 64-bit exist: not restated */
	if(db->v_index >= 0xFFFFFFFFull) goto fail;    /* 64-bit value_index: not restated */
	if(!(db->exist = slurp(f, db->size * 4))) goto fail;
	if(db->DB_size < 65535) {
		if(!(db->values16 = slurp(f, db->v_index * 2))) goto fail;
	} else {
		if(!(db->values32 = slurp(f, db->v_index * 4))) goto fail;
	}
	if(db->mlen <= 16) {
		if(!(db->key32 = slurp(f, (db->n + 1) * 4))) goto fail;
	} else {
		if(!(db->key64 = slurp(f, (db->n + 1) * 8))) goto fail;
	}
	if(!(db->value_index = slurp(f, db->n * 4))) goto fail;
	db->size -= 1; /* becomes the mask, hashmapkma.c:439 */
	if(fread(&db->kmersize, 4, 1, f) == 1) {
		if(fread(&db->flag, 4, 1, f) != 1) goto fail;
	} else {
		db->kmersize = db->mlen; db->flag = 0;
	}
	fclose(f); f = 0;
	if(db->flag) goto fail; /* minimizer / homopolymer indexes are out of scope */

	snprintf(path, sizeof path, "%s.length.b", prefix);
	if((f = fopen(path, "rb"))) {
		int32_t n;
		if(fread(&n, 4, 1, f) != 1 || (uint32_t) n != db->DB_size) goto fail;
		if(!(db->tlen = slurp(f, (size_t) n * 4))) goto fail;
		fclose(f); f = 0;
		db->tseq_off = malloc(((size_t) n + 1) * 8);
		db->tseq_off[0] = 0; db->tseq_off[1] = 0;
		for(int i = 2; i <= n; ++i) {
			db->tseq_off[i] = db->tseq_off[i - 1] + (db->tlen[i - 1] >> 5) + 1;
		}
		snprintf(path, sizeof path, "%s.seq.b", prefix);
		if((f = fopen(path, "rb"))) {
			/* one pad word so getKmer-style reads past the end stay in bounds */
			size_t words = (size_t) db->tseq_off[n];
			db->tseq = calloc(words + 2, 8);
			if(fread(db->tseq, 8, words, f) != words) goto fail;
			fclose(f); f = 0;
		}
	}
	return db;
fail:
	if(f) fclose(f);
	orc_db_free(db);
	return 0;
}

void orc_db_free(orc_db *db) {
	if(!db) return;
	free(db->exist); free(db->key32); free(db->key64); free(db->value_index);
	free(db->values16); free(db->values32); free(db->tlen); free(db->tseq);
	free(db->tseq_off); free(db);
}

/* hashmapkma.c:149-178 with flag == 0: bucket = key & mask; walk the
 * contiguous key run until the key matches or a key of another bucket (the
 * sentinel at [n] is from a different bucket, compress.c:549-585) shows up. */
int64_t orc_hash_get(const orc_db *db, uint64_t key) {
	uint64_t bucket = key & db->size;
	uint64_t pos = db->exist[bucket];
	if(pos == db->null_index) return -1;
	for(;; ++pos) {
		uint64_t k = db->key32 ? db->key32[pos] : db->key64[pos];
		if(k == key) return db->value_index[pos];
		if((k & db->size) != bucket) return -1;
	}
}

const uint64_t *orc_db_template(const orc_db *db, int t) { return db->tseq + db->tseq_off[t]; }
