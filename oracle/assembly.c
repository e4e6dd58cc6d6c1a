/* assembly.c -- TEST INFRASTRUCTURE (see kma_oracle.h).
 *
 * CPU restatement of the per-template half of stage 3c of KMA 1.5.1: the pile-up of the aligned reads
 * (alnToMat, assembly.c:1317-1444, the default sparse matrix with insertion columns chained between template
 * positions), the consensus call (callConsensus :1499-1631, baseCaller :162-179, significantNuc :143-145) and the
 * figures runKMA turns into the identity / coverage / depth columns of a `.res` row (runkma.c:792-809).
 */
#include "kma_oracle.h"
#include <ctype.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { unsigned short counts[6]; int next; } node;

struct orc_assembly {
	node *a;
	int len, size, t_len;
	long score;
};

static inline int tnuc(const uint64_t *s, int pos) { return (int) ((s[pos >> 5] << ((pos & 31) << 1)) >> 62); }

/* assemble_KMA's matrix initialisation, assembly.c:1812-1856: t_len nodes in a ring */
orc_assembly *orc_assembly_new(int t_len) {
	orc_assembly *m = calloc(1, sizeof *m);
	m->t_len = t_len; m->len = t_len; m->size = (t_len << 1) + 2;
	m->a = calloc((size_t) m->size, sizeof(node));
	for(int i = 0; i < t_len; ++i) m->a[i].next = i + 1;
	if(t_len) m->a[t_len - 1].next = 0;
	return m;
}

void orc_assembly_free(orc_assembly *m) { if(m) { free(m->a); free(m); } }

static void bump(unsigned short *c) { if(!++*c) *c = USHRT_MAX; }

/* alnToMat, assembly.c:1317-1444. cols: one char per alignment column, '=' 'X' (aligned pair), 'I' (gap in the template),
 * 'D' (gap in the read); read: the oriented read bytes (0-4) from the first aligned base on (i.e. past the soft clip);
 * start = alnStat.pos. */
void orc_assembly_add(orc_assembly *m, const char *cols, int aln_len, const uint8_t *read, int start, int score) {
	node *A = m->a;
	const int t_len = m->t_len;
	m->score += score;
	/* per column: query code (5 = gap) */
	uint8_t *q = malloc((size_t) aln_len + 1);
	for(int i = 0, r = 0; i < aln_len; ++i) q[i] = cols[i] == 'D' ? 5 : read[r++];
	int i = aln_len - 1;
	while(i && (cols[i] == 'I' || cols[i] == 'D')) --i;           /* trim trailing gaps */
	aln_len = i + 1;
	i = 0;
	while(i < aln_len && (cols[i] == 'I' || cols[i] == 'D')) {     /* trim leading gaps */
		if(cols[i] == 'D') ++start;
		++i;
	}
	int pos = start;
	while(i < aln_len) {
		if(cols[i] == 'I') {
			if(t_len <= pos) {                                       /* an insertion column that already exists */
				bump(&A[pos].counts[q[i]]);
				++i;
				pos = A[pos].next;
			} else {
				const int gaps = pos;
				pos = pos ? (pos - 1) : (t_len - 1);
				while(A[pos].next != gaps) pos = A[pos].next;        /* last column before `gaps` */
				int myBias = 0, tmp = 0;
				for(int j = 0; j < 6; ++j) { myBias += A[pos].counts[j]; tmp += A[gaps].counts[j]; }
				myBias = (tmp < myBias) ? tmp : (myBias - 1);
				if(USHRT_MAX < myBias) myBias = USHRT_MAX;
				while(i < aln_len && cols[i] == 'I') {
					A[pos].next = m->len++;
					if(m->len == m->size) {
						m->size <<= 1;
						m->a = A = realloc(A, (size_t) m->size * sizeof(node));
					}
					pos = A[pos].next;
					A[pos].next = gaps;
					memset(A[pos].counts, 0, sizeof A[pos].counts);
					A[pos].counts[5] = (unsigned short) myBias;
					A[pos].counts[q[i]] = 1;
					++i;
				}
				pos = A[pos].next;
			}
		} else if(t_len <= pos) {                                    /* existing insertion column this read lacks */
			bump(&A[pos].counts[5]);
			pos = A[pos].next;
		} else {
			bump(&A[pos].counts[q[i]]);
			++i;
			pos = A[pos].next;
		}
	}
	free(q);
}

/* significantNuc, assembly.c:143-145 */
static int significant(int X, int Y, double evalue) {
	return (Y < X && orc_p_chisqr(pow(X - Y, 2) / (X + Y)) <= evalue);
}

/* callConsensus with baseCaller, assembly.c:1499-1631, 162-179. Walks the columns in ring order from template position 0.
 * out[0] = cover (consensus base == template base), out[1] = aln_len (called columns), out[2] = depth (sum of the column
 * depths of the called columns), out[3] = number of columns walked. cons (may be NULL, capacity >= len + 1) receives the
 * consensus line ("ACGTN-", lower case = not significant). */
/* significantAnd90Nuc, assembly.c:147-149 */
static int significant90(int X, int Y, double evalue) {
	return (Y < X && (9 * (X + Y) <= 10 * X) && orc_p_chisqr(pow(X - Y, 2) / (X + Y)) <= evalue);
}

void orc_assembly_call(const orc_assembly *m, const uint64_t *tseq, int bcd, double evalue, int64_t *out, char *cons) {
	orc_assembly_call2(m, tseq, bcd, evalue, 0, 0, out, cons);
}

void orc_assembly_call2(const orc_assembly *m, const uint64_t *tseq, int bcd, double evalue, int caller, int sig, int64_t *out, char *cons) {
	static const char bases[] = "ACGTN-";
	const node *A = m->a;
	const int t_len = m->t_len, asm_len = m->len;
	int64_t depth = 0, aln_len = 0, cover = 0;
	int pos = 0;
	for(int i = 0; i < asm_len; ++i) {
		int bestNuc = pos < t_len ? tnuc(tseq, pos) : 5;
		const char tch = bases[bestNuc];
		int bestScore = A[pos].counts[bestNuc];
		long depthUpdate = 0;
		for(int j = 0; j < 6; ++j) {
			if(bestScore < A[pos].counts[j]) { bestScore = A[pos].counts[j]; bestNuc = j; }
			depthUpdate += A[pos].counts[j];
		}
		unsigned char call = (unsigned char) bases[bestNuc];
		if(!depthUpdate) call = '-';
		else if(((long) bestScore << 1) < depthUpdate) {
			if(call == '-') {
				int bestBaseScore = A[pos].counts[4], b = 4;
				for(int j = 0; j < 4; ++j) if(bestBaseScore < A[pos].counts[j]) { bestBaseScore = A[pos].counts[j]; b = j; }
				call = (unsigned char) tolower(bases[b]);
			} else call = (unsigned char) tolower(call);
			bestScore = (int) (depthUpdate - A[pos].counts[5]);
		} else if(depthUpdate < bcd) call = (unsigned char) tolower(call);
		/* baseCaller */
		if(depthUpdate == 0) call = '-';
		else if((sig ? significant90(bestScore, (int) depthUpdate - bestScore, evalue) : significant(bestScore, (int) depthUpdate - bestScore, evalue)) == 0) {
			if(call == '-' && tch != '-' && bestScore != depthUpdate) {
				if(caller == 1) {
					/* nanoCaller, assembly.c:215-230: the best non-gap count decides */
					int bestBaseScore = 0, b = -1;
					for(int j = 0; j < 5; ++j) if(bestBaseScore < A[pos].counts[j]) { bestBaseScore = A[pos].counts[j]; b = j; }
					call = bestBaseScore == 0 ? '-' : (unsigned char) tolower(bases[b]);
				} else call = 'n';
			} else call = (unsigned char) tolower(call);
		}
		if(cons) cons[i] = (char) call;
		if(call != '-') {
			depth += depthUpdate;
			++aln_len;
			if(pos < t_len && tch == toupper(call)) ++cover;
		}
		pos = A[pos].next;
	}
	if(cons) cons[asm_len] = 0;
	out[0] = cover; out[1] = aln_len; out[2] = depth; out[3] = asm_len;
}

int orc_assembly_has_reads(const orc_assembly *m) { return m->score != 0; }
