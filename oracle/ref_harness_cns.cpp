// ref_harness_cns.cpp — TEST INFRASTRUCTURE: function-level harness around the UNMODIFIED mecat2cns aligner
// (reference src/mecat2cns/dw.cpp, compiled where it lies; see oracle/Makefile target `ref`).  Used to pin the C restatement
// orc_cns_* (oracle/mecat_oracle.c) and to generate tests/golden/cns_kats.npz.  Never linked by the product path.
#include <string.h>

#include "mecat2cns/dw.h"

using namespace ns_banded_sw;

static DiffRunningData* g_drd = NULL;
static M5Record* g_m5 = NULL;

static void init_once() {
    if (g_drd) return;
    g_drd = new DiffRunningData(get_sw_parameters_small());      // mecat_correction.cpp: drd_s
    g_m5 = NewM5Record(100000);
}

extern "C" {

// ns_banded_sw::dw (dw.cpp:378-480): res = {query_start, query_end, target_start, target_end, out_store_size, mat, mis, ins, del};
// out1/out2 receive the merged alignment strings ("ACGT-"), NUL terminated
int refc_dw(const char* q, int qstart, int qsize, const char* t, int tstart, int tsize, double error_rate, int min_aln, int* res,
            char* out1, char* out2) {
    init_once();
    OutputStore* r = g_drd->result;
    int ok = dw(q, qsize, qstart, t, tsize, tstart, g_drd->DynQ, g_drd->DynT, g_drd->align, g_drd->d_path, g_drd->aln_path, r, &g_drd->swp,
                error_rate, min_aln);
    res[0] = r->query_start; res[1] = r->query_end; res[2] = r->target_start; res[3] = r->target_end; res[4] = r->out_store_size;
    res[5] = ok ? r->mat : 0; res[6] = ok ? r->mis : 0; res[7] = ok ? r->ins : 0; res[8] = ok ? r->del : 0;
    if (out1) { memcpy(out1, r->out_store1, r->out_store_size); out1[r->out_store_size] = 0; }
    if (out2) { memcpy(out2, r->out_store2, r->out_store_size); out2[r->out_store_size] = 0; }
    return ok;
}

// ns_banded_sw::GetAlignment (dw.cpp:482-553): res = {qoff, qend, soff, send, aligned string length}; qaln/saln as in the M5 record
int refc_get_alignment(const char* q, int qstart, int qsize, const char* t, int tstart, int tsize, double error_rate, int min_aln,
                       int* res, char* qaln, char* saln) {
    init_once();
    bool ok = GetAlignment(q, qstart, qsize, t, tstart, tsize, g_drd, *g_m5, error_rate, min_aln);
    if (!ok) { res[0] = res[1] = res[2] = res[3] = res[4] = 0; return 0; }
    res[0] = (int)m5qoff(*g_m5); res[1] = (int)m5qend(*g_m5); res[2] = (int)m5soff(*g_m5); res[3] = (int)m5send(*g_m5);
    res[4] = (int)strlen(m5qaln(*g_m5));
    if (qaln) strcpy(qaln, m5qaln(*g_m5));
    if (saln) strcpy(saln, m5saln(*g_m5));
    return 1;
}

}  // extern "C"
