/* oracle/drop_attach.c -- TEST INFRASTRUCTURE (built by `make -C oracle drop` into oracle/_ref/sdpasolve_mi355x).
 *
 * The reference-side glue of INTEGRATION.md section 2(b), injected WITHOUT touching the reference: the executable defines
 * HConePresolveData itself.  Calls to that function from inside libhdsdp_ref_minus.so (interface/hdsdp.c:662) go through
 * the PLT, so they land here; we run the reference's own presolve through dlsym(RTLD_NEXT, ...) and then hand every
 * dense SDP block to the engine: the block's CSC user data goes to HMiConeCreateSDP, and the cone's per-iteration slots
 * are re-pointed at the engine's object.  From then on the reference's unmodified driver -- Phase A / Phase B, correctors,
 * line searches, primal recovery -- runs its whole conic work on the GPU through the reference's own cone interface
 * (interface/def_hdsdp_conic.h:60-105).  Two slots keep the CPU cone: getstat (feature detection reads the CPU cone's
 * presolve tables once, interface/hdsdp.c:163) through a trampoline that restores the CPU cone data, and coneView.
 *
 * HDSDP_DROP_ATTACH=0 in the environment leaves the CPU cones in place (then only HKKT* / HFpLinsys* are the engine's).
 * This file contains none of the reference's source; it includes its headers from where they lie, like ref_dump.c. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "interface/hdsdp.h"
#include "interface/hdsdp_utils.h"
#include "interface/def_hdsdp_conic.h"
#include "interface/hdsdp_user_data.h"
#include "interface/def_hdsdp_user_data.h"

/* the engine's constructor (include/hdsdp_mi355x.h; its hdsdp_cone is binary-compatible with the reference's) */
extern hdsdp_retcode HMiConeCreateSDP(hdsdp_cone **pCone, int iCone, int nRow, int nCol, const int *coneMatBeg,
                                      const int *coneMatIdx, const double *coneMatElem, int rank, int world);

/* a crash inside the driver (the reference aborts in a few places, e.g. an undersized dsyevr work array) should say where */
static void crash_backtrace(int sig) {
    void *frames[48];
    const int nf = backtrace(frames, 48);
    const char msg[] = "drop_attach: fatal signal, backtrace:\n";
    (void) !write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, nf, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void install_crash_handler(void) {
    signal(SIGABRT, crash_backtrace);
    signal(SIGSEGV, crash_backtrace);
}

#define MAX_ATTACHED 256
static struct { void *engineData; void *cpuData; void (*cpuGetstat)(void *, double *, int[20], double[20]); } g_tab[MAX_ATTACHED];
static int g_ntab = 0;

static void getstat_trampoline(void *coneData, double *rowRHS, int intFeatures[20], double dblFeatures[20]) {
    for (int i = 0; i < g_ntab; ++i)
        if (g_tab[i].engineData == coneData) { g_tab[i].cpuGetstat(g_tab[i].cpuData, rowRHS, intFeatures, dblFeatures); return; }
}
static void view_nothing(void *coneData) { (void) coneData; }

/* timing only: the reference's HConeProcData (its CPU cone copies and classifies the user data, interface/hdsdp_conic_sdp.c:
   1356-1409) runs before HConePresolveData and is part of what its driver prints as "Pre-solver" time */
hdsdp_retcode HConeProcData(hdsdp_cone *HCone) {
    static hdsdp_retcode (*real)(hdsdp_cone *) = NULL;
    if (!real) real = (hdsdp_retcode (*)(hdsdp_cone *)) dlsym(RTLD_NEXT, "HConeProcData");
    if (!real) { fprintf(stderr, "drop_attach: the reference's HConeProcData was not found\n"); return HDSDP_RETCODE_FAILED; }
    const char *sw = getenv("HDSDP_DROP_ATTACH");
    if (sw && atoi(sw) == 2 && (HCone->cone == HDSDP_CONETYPE_DENSE_SDP || HCone->cone == HDSDP_CONETYPE_SPARSE_SDP)) {
        /* second variant of the glue: the block will live on the engine alone -- its getstat and view slots
           included -- so the reference's CPU cone is never built: no copy of the user data into sdp_coeff objects, no second
           rank-one detection, no dense buffers */
        fprintf(stderr, "drop_attach: cone %d: the reference's HConeProcData is skipped (HDSDP_DROP_ATTACH=2)\n", HCone->iCone);
        return HDSDP_RETCODE_OK;
    }
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    hdsdp_retcode rc = real(HCone);
    clock_gettime(CLOCK_MONOTONIC, &b);
    fprintf(stderr, "drop_attach: cone %d: the reference's own HConeProcData on its CPU cone took %.3f s\n", HCone->iCone,
            (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec));
    return rc;
}

hdsdp_retcode HConePresolveData(hdsdp_cone *HCone) {
    static hdsdp_retcode (*real)(hdsdp_cone *) = NULL;
    if (!real) real = (hdsdp_retcode (*)(hdsdp_cone *)) dlsym(RTLD_NEXT, "HConePresolveData");
    if (!real) { fprintf(stderr, "drop_attach: the reference's HConePresolveData was not found\n"); return HDSDP_RETCODE_FAILED; }
    struct timespec ts0, ts1, ts2;
    const char *sw = getenv("HDSDP_DROP_ATTACH");
    const int engine_only = sw && atoi(sw) == 2 && (HCone->cone == HDSDP_CONETYPE_DENSE_SDP || HCone->cone == HDSDP_CONETYPE_SPARSE_SDP);
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    hdsdp_retcode rc = engine_only ? HDSDP_RETCODE_OK : real(HCone);
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    if (rc != HDSDP_RETCODE_OK) return rc;
    /* dense SDP cones, and the reference's sparse SDP cones (blocks on which most constraints are zero,
       hdsdp_conic_sdp.c:1814-1886): the engine keeps only the constraints that have data on a block */
    if ((sw && atoi(sw) == 0) || (HCone->cone != HDSDP_CONETYPE_DENSE_SDP && HCone->cone != HDSDP_CONETYPE_SPARSE_SDP) ||
        g_ntab >= MAX_ATTACHED) return rc;
    user_data *u = (user_data *) HCone->usrData;
    hdsdp_cone *g = NULL;
    rc = HMiConeCreateSDP(&g, HCone->iCone, u->nConicRow, u->nConicCol, u->coneMatBeg, u->coneMatIdx, u->coneMatElem, 0, 1);
    if (rc != HDSDP_RETCODE_OK) { fprintf(stderr, "drop_attach: HMiConeCreateSDP failed for cone %d\n", HCone->iCone); return rc; }
    g_tab[g_ntab].engineData = g->coneData;
    g_tab[g_ntab].cpuData = HCone->coneData;
    g_tab[g_ntab].cpuGetstat = HCone->getstat;
    g_ntab += 1;
    HCone->coneData = g->coneData;                 /* the CPU cone's data stays alive for getstat (and is leaked at exit) */
    HCone->coneDestroyData = g->coneDestroyData;
    HCone->coneSetStart = g->coneSetStart;                     HCone->coneUpdate = g->coneUpdate;
    HCone->coneRatioTest = g->coneRatioTest;
    HCone->coneGetSymNnz = g->coneGetSymNnz;                   HCone->coneGetDim = g->coneGetDim;
    HCone->coneAddSymNz = g->coneAddSymNz;                     HCone->coneGetKKTMap = g->coneGetKKTMap;   /* sparse Schur pattern */
    HCone->coneBuildSchur = g->coneBuildSchur;                 HCone->coneBuildSchurFixed = g->coneBuildSchurFixed;
    HCone->coneBuildPrimalDirection = g->coneBuildPrimalDirection;
    HCone->coneInteriorCheck = g->coneInteriorCheck;           HCone->coneInteriorCheckExpert = g->coneInteriorCheckExpert;
    HCone->coneGetBarrier = g->coneGetBarrier;                 HCone->coneAxpyBufferAndCheck = g->coneAxpyBufferAndCheck;
    HCone->coneReduceResi = g->coneReduceResi;                 HCone->coneSetPerturb = g->coneSetPerturb;
    HCone->conePRecover = g->conePRecover;                     HCone->coneDRecover = g->coneDRecover;
    HCone->coneATimesXpy = g->coneATimesXpy;                   HCone->coneTraceCX = g->coneTraceCX;
    HCone->coneXDotS = g->coneXDotS;                           HCone->coneGetCoeffNorm = g->coneGetCoeffNorm;
    HCone->coneGetObjNorm = g->coneGetObjNorm;                 HCone->coneScal = g->coneScal;
    if (engine_only) { HCone->getstat = g->getstat; HCone->coneView = g->coneView; }   /* the engine answers from its own presolve */
    else { HCone->getstat = getstat_trampoline; HCone->coneView = view_nothing; }
    clock_gettime(CLOCK_MONOTONIC, &ts2);
    fprintf(stderr, "drop_attach: cone %d (n = %d, m = %d) attached to the MI355X engine (the reference's own HConePresolveData on its CPU cone "
                    "%.3f s, HMiConeCreateSDP %.3f s)\n", HCone->iCone, u->nConicCol, u->nConicRow,
            (ts1.tv_sec - ts0.tv_sec) + 1e-9 * (ts1.tv_nsec - ts0.tv_nsec), (ts2.tv_sec - ts1.tv_sec) + 1e-9 * (ts2.tv_nsec - ts1.tv_nsec));
    free(g);                                       /* the shell only carried the slots */
    return HDSDP_RETCODE_OK;
}
