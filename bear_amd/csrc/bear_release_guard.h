// bear_release_guard.h -- a release build must not be compiled with a developer switch defined.
//
// Until round 6 the kernel sources carried "timing only, results meaningless" branches (fake LDS rows, skipped phases, dropped
// barriers) behind macros: one stray -D and the shipped library computed garbage.  Those branches now live in
// scripts/dev/patches/timing_switches.patch; this header makes sure that neither their names nor the stamp / probe switches that
// remain in the sources (developer builds that add clock read-outs to the kernels) are defined unless the build says it is a
// developer build (-DBEAR_DEV_BUILD: never what `make` builds).
#pragma once
#ifndef BEAR_DEV_BUILD
#if defined(LIN_FAKE_SHARED_ROWS) || defined(LIN_FAKE_T_ROWS) || defined(LIN_FAKE_PRI_ROWS) || defined(LIN_FAKE_TRIPLE_ROWS) ||      \
    defined(LIN_FAKE_ITEM_OFFS) || defined(LIN_SKIP_A_ROWS) || defined(LIN_SKIP_A) || defined(LIN_SKIP_B) || defined(LIN_SKIP_C) ||   \
    defined(LIN_SKIP_TRIPLE) || defined(LIN_SKIP_PAIRS3) || defined(LIN_SKIP_SCATTER) || defined(LIN_MIX) || defined(LIN_NOSYNC) ||   \
    defined(LIN_DBG) || defined(LIN_ONLY_EXP) || defined(PLN_NOWORK) || defined(EVP_DEBUG_SWITCHES)
#error "a timing-only developer switch is defined (results would be meaningless): release builds refuse it -- see scripts/dev/patches/timing_switches.patch"
#endif
#if defined(LIN_STAMPS) || defined(PLN_STAMPS) || defined(EVP_STAMPS) || defined(CNN_STAMPS) || defined(PLN_NO_FIRST_UNIT_DEALT) || \
    defined(EVP_TICKET_PREFETCH) || defined(CNN_NO_SHARED_BACKWARD) || defined(CNN_NO_SKIP)
#error "a developer measurement switch is defined without -DBEAR_DEV_BUILD: release builds refuse it"
#endif
#endif
