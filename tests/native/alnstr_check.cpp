// self-check of mecat_amd/csrc/aln_strings.h (the accept stage's string builder, four columns at a time) against its column-by-column form:
// random column sets (ragged lengths, junk behind the last column of a word), both parts.  Built and run by tests/test_alnstr_cpu.py.
#include "aln_strings.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string>
int main() {
    srand(5);
    for (int it = 0; it < 20000; ++it) {
        const int L = rand() % 200, R = rand() % 200;
        std::vector<uint32_t> lw((L + 15) / 16 + 1), rw((R + 15) / 16 + 1);
        auto fill = [&](std::vector<uint32_t>& w, int n) { for (auto& x : w) x = 0; for (int k = 0; k < n; ++k) { int op = rand() % 10; op = op < 7 ? 0 : op < 9 ? 1 : 2; w[k >> 4] |= (uint32_t)op << ((k & 15) << 1); }
            // junk beyond n in the last word
            for (int k = n; k < (int)w.size() * 16; ++k) w[k >> 4] |= (uint32_t)(rand() & 3) << ((k & 15) << 1); };
        fill(lw, L); fill(rw, R);
        std::string qs(8 + L + R + 8, 'x'), ts(8 + L + R + 8, 'y');
        for (auto& c : qs) c = "ACGT"[rand() & 3];
        for (auto& c : ts) c = "ACGT"[rand() & 3];
        std::string a(8 + L + R + 8, '#'), b(a), c(a), d(a);
        alnstr::build(lw.data(), L, rw.data(), R, qs.data() + 8, ts.data() + 8, &a[8], &b[8]);
        alnstr::build_plain(lw.data(), L, rw.data(), R, qs.data() + 8, ts.data() + 8, &c[8], &d[8]);
        if (a.substr(8, L + R) != c.substr(8, L + R) || b.substr(8, L + R) != d.substr(8, L + R)) { printf("MISMATCH it %d L %d R %d\n%s\n%s\n", it, L, R, a.c_str(), c.c_str()); return 1; }
    }
    printf("ok\n");
}
