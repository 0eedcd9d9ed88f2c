// cnf_tuning.hip - the library's ONE switchboard (include/cnf.h: cnf_tuning): defaults, the CNF_* environment variables that
// override them at the first use of the board (and again on cnf_set_tuning(NULL)), cnf_get_tuning / cnf_set_tuning.  No other file of the library reads the environment.
#include <atomic>
#include <cstdlib>
#include <mutex>

#include "cnf_internal.h"

namespace cnf {

namespace {
struct Knob { const char* env; int32_t cnf_tuning::*field; int32_t dflt; };
const Knob kKnobs[] = {
    {"CNF_TILE_SPLIT", &cnf_tuning::tile_split, 1},
    {"CNF_SOLVE2", &cnf_tuning::solve2, 2},
    {"CNF_SOLVE2_PAIR", &cnf_tuning::solve2_pair, 1},
    {"CNF_COOPD", &cnf_tuning::coopd, 1},
    {"CNF_COOPD_GRAD", &cnf_tuning::coopd_grad, 1},
    {"CNF_COOP_GRAD", &cnf_tuning::coop_grad, 1},
    {"CNF_COOP_GRAD_MID", &cnf_tuning::coop_grad_mid, 1},
    {"CNF_COOP_GRAD3", &cnf_tuning::coop_grad3, 1},
    {"CNF_COOP_GRAD3_GIB", &cnf_tuning::coop_grad3_gib, 96},
    {"CNF_GRAD_LAYERED", &cnf_tuning::grad_layered, 0},
    {"CNF_JVP_GRAD_TWIN", &cnf_tuning::jvp_grad_twin, 1},
    {"CNF_PROBE_GRAD_TWIN", &cnf_tuning::probe_grad_twin, 1},
    {"CNF_ADAPTIVE_CKPT", &cnf_tuning::adaptive_ckpt, 1},
    {"CNF_LAYERED_LOSS_BY_SOLVE", &cnf_tuning::layered_loss_by_solve, 0},
    {"CNF_DEVICE_CONTROLLER", &cnf_tuning::device_controller, 1},
    {"CNF_DC_PER_CU", &cnf_tuning::dc_per_cu, 2},
    {"CNF_MFMA_COOP", &cnf_tuning::mfma_coop, 0},
    {"CNF_MFMA_COOPX", &cnf_tuning::mfma_coopx, 1},
    {"CNF_MFMA_NT", &cnf_tuning::mfma_nt, 0},
    {"CNF_MFMA_PRE", &cnf_tuning::mfma_pre, -1},
    {"CNF_MFMA_PRIO", &cnf_tuning::mfma_prio, 0},
    {"CNF_MFMA_QUEUE", &cnf_tuning::mfma_queue, 0},
    {"CNF_CG_ONE_PER_CU", &cnf_tuning::cg_one_per_cu, 0},
    {"CNF_LAYERED_MIN_B", &cnf_tuning::layered_min_b, 0},
    {"CNF_LAYERED_KC", &cnf_tuning::layered_kc, 0},
    {"CNF_LAYERED_NO_KCKPT", &cnf_tuning::layered_no_kckpt, 0},
    {"CNF_LAYERED_ACT_GIB", &cnf_tuning::layered_act_gib, 48},
    {"CNF_LG_GEMM", &cnf_tuning::lg_gemm, 2},
    {"CNF_LG_SPW", &cnf_tuning::lg_spw, 0},
    {"CNF_LG_NW", &cnf_tuning::lg_nw, 4},
    {"CNF_LG_GEMM2_WIDE", &cnf_tuning::lg_gemm2_wide, 1},
    {"CNF_LG_WGRAD_PER_CU", &cnf_tuning::lg_wgrad_per_cu, 0},
    {"CNF_LG_WGRAD_T1", &cnf_tuning::lg_wgrad_t1, 5},
    {"CNF_LG_WGRAD_T2", &cnf_tuning::lg_wgrad_t2, 8},
};
cnf_tuning board_defaults() {
    cnf_tuning t{};
    for (const Knob& k : kKnobs) t.*(k.field) = k.dflt;
    return t;
}
// defaults, then every CNF_* variable that is set and not empty (integers; "1"-style flags read as their number)
cnf_tuning board_from_env() {
    cnf_tuning t = board_defaults();
    for (const Knob& k : kKnobs) {
        const char* v = getenv(k.env);
        if (v && *v) t.*(k.field) = (int32_t)atoi(v);
    }
    return t;
}
// The board is published as an immutable snapshot behind one atomic pointer: a reader - any call of the library, on
// any thread, on any handle - sees one consistent board for as long as it holds the reference, and a writer never edits a board
// somebody reads.  Superseded snapshots are not freed (a reader may still hold one; ~150 bytes per change).
std::atomic<const cnf_tuning*> g_board{nullptr};
std::mutex g_board_mu;
std::once_flag g_board_once;
void publish(const cnf_tuning& t) {
    std::lock_guard<std::mutex> lk(g_board_mu);
    g_board.store(new cnf_tuning(t), std::memory_order_release);
}
}  // namespace

// The environment is read ONCE, at the first use of the board, and again only on request (cnf_set_tuning(NULL)): creating a
// handle - the library itself creates internal ones - never reverts what cnf_set_tuning set.
const cnf_tuning& tuning() {
    std::call_once(g_board_once, [] { publish(board_from_env()); });
    return *g_board.load(std::memory_order_acquire);
}

void tuning_from_env() {
    (void)tuning();
    publish(board_from_env());
}

}  // namespace cnf

extern "C" {
int cnf_get_tuning(cnf_tuning* out) {
    if (!out) return CNF_ERR_INVALID;
    *out = cnf::tuning();
    return CNF_OK;
}
int cnf_set_tuning(const cnf_tuning* in) {
    if (!in) cnf::tuning_from_env();   // NULL: the defaults + the CNF_* variables again
    else { (void)cnf::tuning(); cnf::publish(*in); }
    return CNF_OK;
}
}
