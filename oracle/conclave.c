/* conclave.c -- TEST INFRASTRUCTURE (see kma_oracle.h).
 *
 * CPU restatement of stage 3b of KMA 1.5.1: the ConClave template choice per read
 * (runConClave, conclave.c:43-215, the default `-ConClave 1`) and the per-template
 * statistics that open every `.res` row (runkma.c:608-613, 765-783; p_chisqr / fastp,
 * stdstat.c:36-147). Works on arrays (one record per frag_raw record) instead of the
 * reference's temp-file stream.
 */
#include "kma_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* runConClave, conclave.c:59-133: among the templates a read aligned equally well to, take the one with the
 * largest alignment_scores; ties -> larger alignment_scores / template_length (double), -> larger
 * uniq_alignment_scores, -> smaller template id. The reference keeps the running best score and uniq count in
 * `int` variables (conclave.c:45-46) and compares them with the unsigned long vector entries: the truncation
 * and the sign extension of that comparison are kept. Returns the index of the chosen hit. */
static int conclave_pick(int n, const int32_t *tmpl, const uint64_t *as, const uint64_t *us, const int32_t *tlen) {
	int best = -1, best_tmpl = -1, best_read_score = 0, best_num = 0;
	double best_score = 0;
	for(int i = 0; i < n; ++i) {
		const int tt = tmpl[i], t = tt < 0 ? -tt : tt;
		const double sc = 1.0 * as[t] / tlen[t];
		int take = 0;
		if(as[t] > (uint64_t) (int64_t) best_read_score) take = 1;
		else if(as[t] == (uint64_t) (int64_t) best_read_score) {
			if(sc > best_score) take = 1;
			else if(sc == best_score) {
				if(us[t] > (uint64_t) (int64_t) best_num) take = 1;
				else if(us[t] == (uint64_t) (int64_t) best_num && t < abs(best_tmpl)) take = 1;
			}
		}
		if(take) { best = i; best_tmpl = tt; best_read_score = (int) as[t]; best_score = sc; best_num = (int) us[t]; }
	}
	return best;
}

/* One record = one frag_raw record (updatescores.c:283-295, 360-388, 470-488): n_hits templates with their
 * start / end, read_score = |stats[2]| (negative in the record: a second mate follows, conclave.c:171),
 * q_len / q_len2 = lengths of the read and of the mate (0 if none).
 * Out per record: chosen signed template, its start / end; per template: w_scores (conclave.c:147),
 * fragmentCounts / readCounts (:148-151, 172-174) and the summed read lengths (what skip_assemble_KMA turns
 * into Depth, assembly.c:1280). A record with n_hits == 0 and read_score == 0 stands for "no record written";
 * n_hits == 0 with a score is the empty-list record described below. Records must be in stream order.
 * Returns 0, or -1 if a multi-hit record finds no template (the reference would index alignFrags[-1]). */
int orc_conclave(int64_t n_rec, const int32_t *n_hits, const int32_t *read_score, const int32_t *q_len, const int32_t *q_len2,
                 const int64_t *off, const int32_t *tmpl, const int32_t *start, const int32_t *end,
                 const uint64_t *alignment_scores, const uint64_t *uniq_alignment_scores, const int32_t *template_lengths,
                 int32_t *out_tmpl, int32_t *out_start, int32_t *out_end,
                 uint64_t *w_scores, uint32_t *fragmentCounts, uint32_t *readCounts, uint64_t *depth) {
	int stale_t = 0, stale_s = 0, stale_e = 0;
	for(int64_t r = 0; r < n_rec; ++r) {
		const int n = abs(n_hits[r]);
		out_tmpl[r] = 0; out_start[r] = out_end[r] = 0;
		if(n == 0 && read_score[r] == 0) continue;      /* nothing was written for this read */
		const int64_t o = off[r];
		int tt, st, en;
		if(n > 1) {
			const int pick = conclave_pick(n, tmpl + o, alignment_scores, uniq_alignment_scores, template_lengths);
			if(pick < 0) return -1;
			tt = tmpl[o + pick]; st = start[o + pick]; en = end[o + pick];
		} else if(n == 1) {
			tt = tmpl[o]; st = start[o]; en = end[o];
		} else {
			/* update_Scores_pe can write a record whose hit list is empty (no kept score equals the pair's best,
			 * updatescores.c:402-417): runConClave then reads zero list entries and takes element 0 of its buffers
			 * (conclave.c:123-127), i.e. the first listed hit of the last record that had any. */
			tt = stale_t; st = stale_s; en = stale_e;
		}
		if(n) { stale_t = tmpl[o]; stale_s = start[o]; stale_e = end[o]; }
		const int t = tt < 0 ? -tt : tt;
		out_tmpl[r] = tt; out_start[r] = st; out_end[r] = en;
		if(t == 0) continue;                            /* stale buffer never filled: the reference reads fresh memory */
		w_scores[t] += (uint64_t) abs(read_score[r]);
		if(fragmentCounts) { fragmentCounts[t]++; readCounts[t]++; }
		if(depth) depth[t] += (uint64_t) q_len[r];
		if(read_score[r] < 0) {
			if(readCounts) readCounts[t]++;
			if(depth) depth[t] += (uint64_t) q_len2[r];
		}
	}
	return 0;
}

/* fastp, stdstat.c:36-134: p-value of a chi-square quantile (1 d.o.f.) from a 48-step table */
static double chi2_table_p(long double q) {
	static const double thr[] = {
		114.5242, 109.9604, 105.3969, 100.8337, 96.27476, 91.71701, 87.16164, 82.60901, 78.05917, 73.51245, 68.96954,
		64.43048, 59.89615, 55.36699, 50.84417, 46.32844, 41.82144, 37.32489, 32.84127, 28.37395, 23.92814, 19.51139,
		15.13671, 10.82759, 6.634897, 3.841443, 2.705532, 2.072251, 1.642374, 1.323304, 1.074194, 0.8734571, 0.7083263,
		0.5706519, 0.4549364, 0.3573172, 0.2749959, 0.2059001, 0.1484719, 0.1015310, 0.06418475, 0.03576578, 0.01579077,
		0.00393214 };
	static const double pv[] = {
		1e-26, 1e-25, 1e-24, 1e-23, 1e-22, 1e-21, 1e-20, 1e-19, 1e-18, 1e-17, 1e-16, 1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10,
		1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 0.01, 0.05, 0.1, 0.15, 0.2, 0.25, 0.3, 0.35, 0.4, 0.45, 0.5, 0.55, 0.6, 0.65,
		0.7, 0.75, 0.8, 0.85, 0.9, 0.95 };
	for(size_t i = 0; i < sizeof thr / sizeof thr[0]; ++i) if(q > thr[i]) return pv[i];
	if(q >= 0.0) return 1.0;
	return 1.00 - chi2_table_p(-1 * q);
}

/* p_chisqr, stdstat.c:136-147 */
double orc_p_chisqr(long double q) {
	if(q < 0) return 1e-26;
	if(q > 49) return chi2_table_p(q);
	return 1 - 1.772453850 * erf(sqrt(0.5 * q)) / tgamma(0.5);
}

/* The leading columns of a `.res` row (runkma.c:608-613, 765-783): for every template with w_scores > 0:
 * expected = t_len / max(1, tot_len - t_len) * (Nhits - score); q = (score - expected)^2 / (expected + score);
 * p = p_chisqr(q); significant = cmp_or(p <= evalue && score > expected, score >= scoreT * t_len)
 * (stdstat.c:23-27, the default cmp). expected is returned as the row prints it ((unsigned) expected), q as (double).
 * Arrays are DB_size long; returns the number of templates with a score. */
int orc_res_stats(int DB_size, const uint64_t *w_scores, const int32_t *template_lengths, double evalue, double scoreT,
                  double *expected, double *q_value, double *p_value, int32_t *significant) {
	long unsigned Nhits = 0, tot = 0;
	int rows = 0;
	for(int i = DB_size - 1; i > 0; --i) { tot += template_lengths[i]; Nhits += w_scores[i]; }
	Nhits = Nhits ? Nhits : 1;
	for(int t = 1; t < DB_size; ++t) {
		expected[t] = q_value[t] = 0; p_value[t] = 1; significant[t] = 0;
		if(!(w_scores[t] > 0)) continue;
		++rows;
		const long read_score = (long) w_scores[t];
		const int t_len = template_lengths[t];
		long double e = t_len, q;                 /* runkma.c:141: expected and q_value are long double */
		const long unsigned denom = tot - t_len;
		e /= (1 < denom ? denom : 1);
		e *= (Nhits - read_score);
		if(0 < e) {
			q = read_score - e;
			q /= (e + read_score);
			q *= (read_score - e);
		} else q = read_score;
		const double p = orc_p_chisqr(q);
		expected[t] = (double) (unsigned) e;     /* as printed: "%8u" of (unsigned) expected */
		q_value[t] = (double) q; p_value[t] = p;
		significant[t] = ((p <= evalue && read_score > e) || (read_score >= scoreT * t_len)) ? 1 : 0;
	}
	return rows;
}
