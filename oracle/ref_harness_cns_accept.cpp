// ref_harness_cns_accept.cpp — TEST INFRASTRUCTURE: runs the UNMODIFIED consensus_one_read_can_pacbio / _nanopore of the
// reference (src/mecat2cns/mecat_correction.cpp:388-515, compiled where it lies together with the rest of mecat2cns; see
// oracle/Makefile target `ref`) on one template and reports what it accepted: the entries it left in ConsensusThreadData::cns_alns
// (CnsAlns::add_aln, reads_correction_aux.h:87-99) — template end coordinate, aligned string length and the two gap-normalised
// strings.  consensus_worker advances CnsAln::soff while it reads the alignments, so soff is re-derived as
// send - (non-gap characters of saln).  Pins mhip_cns_accept_templates; never linked by the product path.
#include <string.h>

#include <fstream>
#include <sstream>

#include "mecat2cns/mecat_correction.h"
#include "mecat2cns/reads_correction_aux.h"

using namespace ns_meap_cns;

static PackedDB* g_reads = NULL;
static ConsensusThreadData* g_ctd = NULL;
static std::ostringstream g_sink;

extern "C" {

// all reads from a FASTA file, as mecat2cns loads them (reads_correction_can.cpp)
int refa_load_reads(const char* fasta) {
    delete g_reads;
    g_reads = new PackedDB();
    g_reads->load_fasta_db(fasta);
    return (int)g_reads->num_seqs();
}

// candidates: n x 13 ints in ExtensionCandidate field order, all of ONE template (sid == read_id); sorted in place by the reference.
// Returns the number of accepted alignments; out_meta[4 * i ..] = {soff, send, aln_size, qid-as-found (-1)}, strings appended to
// out_strings as qaln NUL saln NUL.
int refa_consensus_can(int tech, int* cands, int n, int read_id, int min_align_size, double min_mapping_ratio, int* out_meta,
                       char* out_strings, long out_cap, long* out_used) {
    ReadsCorrectionOptions rco;
    memset(&rco, 0, sizeof(rco));
    rco.input_type = INPUT_TYPE_CAN;
    rco.num_threads = 1;
    rco.min_mapping_ratio = min_mapping_ratio;
    rco.min_align_size = min_align_size;
    rco.min_cov = tech == TECH_PACBIO ? 4 : 6;         // mecat2cns defaults (options.cpp); they only steer consensus_worker
    rco.min_size = tech == TECH_PACBIO ? 5000 : 2000;
    rco.tech = tech;
    delete g_ctd;
    g_ctd = new ConsensusThreadData(&rco, 0, g_reads, (ExtensionCandidate*)cands, n, &g_sink);
    if (tech == TECH_PACBIO) consensus_one_read_can_pacbio(g_ctd, read_id, 0, n);
    else consensus_one_read_can_nanopore(g_ctd, read_id, 0, n);
    long used = 0;
    int k = 0;
    for (CnsAln* a = g_ctd->cns_alns.begin(); a != g_ctd->cns_alns.end(); ++a, ++k) {
        int tb = 0;
        for (int i = 0; i < a->aln_size; ++i) tb += a->saln[i] != '-';
        out_meta[4 * k] = a->send - tb;
        out_meta[4 * k + 1] = a->send;
        out_meta[4 * k + 2] = a->aln_size;
        out_meta[4 * k + 3] = -1;
        if (used + 2L * (a->aln_size + 1) > out_cap) return -1;
        memcpy(out_strings + used, a->qaln, (size_t)a->aln_size + 1);
        used += a->aln_size + 1;
        memcpy(out_strings + used, a->saln, (size_t)a->aln_size + 1);
        used += a->aln_size + 1;
    }
    *out_used = used;
    g_ctd->cns_results.clear();
    return k;
}

// the reference's normalize_gaps(.., push = true) on one pair of strings (reads_correction_aux.cpp:3-81): pins the device kernel that
// pushes the gaps of the accepted alignments (cns_strings.hip) on adversarial gap patterns.  Returns the normalised length.
int refa_normalize_gaps(const char* q, const char* t, int n, char* qout, char* tout, int cap) {
    std::string qn, tn;
    normalize_gaps(q, t, n, qn, tn, true);
    if ((int)qn.size() + 1 > cap) return -1;
    memcpy(qout, qn.c_str(), qn.size() + 1);
    memcpy(tout, tn.c_str(), tn.size() + 1);
    return (int)qn.size();
}

}  // extern "C"
