/*
 * m4sort.cpp — TEST INFRASTRUCTURE (part of the oracle).
 * append_m4v (mecat2pw/pw_impl.cpp:581) orders a read's records with libstdc++ std::sort and
 * CmpM4RecordByQidAndOvlpSize (:539-548).  std::sort is not stable, so the tie order is libstdc++'s introsort order;
 * the only faithful restatement is to call the same routine.  (The reference is built with -D_GLIBCXX_PARALLEL, whose
 * std::sort falls back to this sequential algorithm below 1000 elements; a read has at most MAXC = 100 records.)
 */
#include <algorithm>
#include "mecat_oracle.h"

namespace {
struct CmpByQidAndOvlpSize {
    bool operator()(const orc_m4& a, const orc_m4& b) const {
        if (a.qid != b.qid) return a.qid < b.qid;
        /* M4RecordOverlapSize, common/alignment.h:104-109, truncated to int like the reference's `int o1` */
        int64_t qa = a.qend - a.qoff, sa = a.send - a.soff, qb = b.qend - b.qoff, sb = b.send - b.soff;
        int o1 = (int)(qa <= sa ? qa : sa), o2 = (int)(qb <= sb ? qb : sb);
        return o1 > o2;
    }
};
}
extern "C" void orc_m4_std_sort(orc_m4* list, int n) { std::sort(list, list + n, CmpByQidAndOvlpSize()); }
