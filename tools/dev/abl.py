"""Ablation / variant builds of the library for A/B timing (tools/gpu_mfe_cmp.py picks up every tools/abl_*.so).

    python tools/dev/abl.py NAME [NAME ...]     # builds tools/abl_NAME.so for each, in parallel

A variant is the working tree's scanfold_amd/csrc + include copied to /tmp/abl_NAME with text patches applied (PATCHES below:
(anchor, replacement) pairs; an anchor must occur exactly once) and / or extra -D flags.  Ablated kernels compute WRONG
energies on purpose (a section skipped): only their time and counters mean anything.  Never part of the product."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FAST = "scanfold_amd/csrc/sf_mfe_fast.hip.h"

CELLS = "      if (__ballot(valid)) {\n        if (DO_G && d0 < 8)"
FIN = "        if (!helper && !dmlw && __ballot(valid)) {"
DML2 = "            dec = sf_fast_dml2<WT, LGC>(X, d, tid & 63, i, valid);"
MAINCALL = ("            sf_fast_cell<false, WT, SF_SEC_P1 | SF_SEC_C0 | SF_SEC_PRE, false, FOLD, SFD_MAXLOOP, TBLK, false, UNPK && P2>(X, d, i, valid, "
            "slot2, slotd, H, HU, ovf, grp == 0, fnb, fpart, dec, eh, e0, dprev, pub);")
MGHCALL = "            sf_fast_cell<false, WT, SF_SEC_HELP, false, FOLD, SFD_MAXLOOP, TBLK, true>(Xh, d0 + g, iC,"

HOST = "scanfold_amd/csrc/scanfold_hip.hip"
# "stamps": a cycle-stamped diagnostic build.  Lane 0 of every wave of every 64th workgroup adds, per step and wave, the cycles
# (s_memtime) from the start of the step to: the end of its own pre-barrier work [0], the exit of the exchange barrier [5] (split
# steps), the end of the finish [1], the exit of the end-of-step barrier [2], the end of the step (after the fML fix-up) [3];
# [4] counts.  The sums live behind the status word; sf_prof_get dumps them to $SF_STAMP_OUT (tools/dev/stamp_report.py reads it).
STAMP_PATCHES = [
    (FAST, "      constexpr bool DO_G = (PH == 0 || PH == 3), DO_CH = (PH == 0 || PH == 4);\n      const int d = d0 + grp;\n",
     "      constexpr bool DO_G = (PH == 0 || PH == 3), DO_CH = (PH == 0 || PH == 4);\n      const int d = d0 + grp;\n"
     "      unsigned long long *const SFP = (unsigned long long *)(status + 64) + (size_t)(((d0 >> 1) * 4 + (tid >> 6)) * 8);\n"
     "      const bool SFL = (tid & 63) == 0 && (blockIdx.x & 63) == 5;\n"
     "      const long long SFT0 = SFL ? (long long)clock64() : 0;\n"),
    (FAST, "      if (split) {\n        __syncthreads();\n        if (!helper && !dmlw && __ballot(valid)) {",
     "      if (SFL) atomicAdd(SFP + 0, (unsigned long long)(clock64() - SFT0));\n"
     "      if (split) {\n        __syncthreads();\n        if (SFL) atomicAdd(SFP + 5, (unsigned long long)(clock64() - SFT0));\n"
     "        if (!helper && !dmlw && __ballot(valid)) {"),
    (FAST, "      __syncthreads();\n// @section fml_fixup\n",
     "      if (SFL) atomicAdd(SFP + 1, (unsigned long long)(clock64() - SFT0));\n      __syncthreads();\n"
     "      if (SFL) atomicAdd(SFP + 2, (unsigned long long)(clock64() - SFT0));\n"),
    (FAST, "      slot2 += 2; if (slot2 >= SF_FAST_NR) slot2 -= SF_FAST_NR;\n      slotd += 2; if (slotd >= SF_FAST_NR) slotd -= SF_FAST_NR;\n    };",
     "      if (SFL) { atomicAdd(SFP + 3, (unsigned long long)(clock64() - SFT0)); atomicAdd(SFP + 4, 1ull); }\n"
     "      slot2 += 2; if (slot2 >= SF_FAST_NR) slot2 -= SF_FAST_NR;\n      slotd += 2; if (slotd >= SF_FAST_NR) slotd -= SF_FAST_NR;\n    };"),
    (FAST, "      if (__ballot(valid)) {\n        if (DO_G && d0 < 8)",
     "      if (SFL) atomicAdd(SFP + 6, (unsigned long long)(clock64() - SFT0));\n      if (__ballot(valid)) {\n        if (DO_G && d0 < 8)"),
    (FAST, "            dec = sf_fast_dml2<WT, LGC>(X, d, tid & 63, i, valid);\n",
     "            dec = sf_fast_dml2<WT, LGC>(X, d, tid & 63, i, valid);\n            SF_PIN(dec);\n"
     "            if (SFL) atomicAdd(SFP + 7, (unsigned long long)(clock64() - SFT0));\n"),
    (HOST, "    int rc = ensure(g.status, sizeof(int));\n    if (rc) return rc;\n    HIPCHK(hipMemset(g.status.p, 0, sizeof(int)));",
     "    int rc = ensure(g.status, 65536);\n    if (rc) return rc;\n    HIPCHK(hipMemset(g.status.p, 0, 65536));"),
    (HOST, "  ProfPair::prof_drain();\n  if (ms) *ms = g.prof_ms;",
     "  ProfPair::prof_drain();\n  if (const char *so = getenv(\"SF_STAMP_OUT\")) {\n"
     "    static unsigned long long hb[64 * 4 * 8];\n    HIPCHK(hipMemcpy(hb, (char *)g.status.p + 256, sizeof hb, hipMemcpyDeviceToHost));\n"
     "    HIPCHK(hipMemset((char *)g.status.p + 256, 0, sizeof hb));\n    if (FILE *f = fopen(so, \"w\")) {\n"
     "      for (int k = 0; k < 64 * 4; k++) { for (int q = 0; q < 8; q++) fprintf(f, \"%llu \", hb[k * 8 + q]); fprintf(f, \"\\n\"); }\n"
     "      fclose(f);\n    }\n  }\n  if (ms) *ms = g.prof_ms;"),
    (HOST, "  g.prof_on = true;  // from now on launch_mfe brackets the dominant kernel with two events",
     "  g.prof_on = true;\n  HIPCHK(hipMemset((char *)g.status.p + 256, 0, 64 * 4 * 8 * 8));"),
]

# "pfstamps": the same for sf_pf_lds_kernel.  Lane 0 of every wave of every 64th workgroup adds, per wave (8 waves x 64 slots of
# cycles behind the status word at byte 32768): outside pass per column, in four buckets of the column l (bucket (l - 1) / 30, eight
# slots each) — a point inside the own work [0] (team 0: after the recurrence batches; teams 1-3: before the R1 share), own work [1],
# exit of the first barrier [2], end of the post-barrier work [3], exit of the second barrier [4], columns [5]; the inside columns
# [32..36] (own, barrier 1, post, barrier 2, columns); per fold (wave 0 only) — reload [40], inside loop [41], park + exterior + tables
# [42], outside loop [43], reductions [44], folds [45].
# launch_pf waits for the kernel and appends the sums to $SF_STAMP_OUT (tools/dev/pf_stamp_report.py).
PFL = "scanfold_amd/csrc/sf_pf_lds.hip.h"
PFSTAMP_PATCHES = [
    (PFL, "    const bool ovl = SF_PFL_NZP * VW >= 900;\n",
     "    const long long SFX0 = SFL ? (long long)clock64() : 0;\n    const bool ovl = SF_PFL_NZP * VW >= 900;\n"),
    (PFL, "    __syncthreads();\n    if (tid >= 128) {\n      const int ht = tid - 128;",
     "    __syncthreads();\n    if (SFL) atomicAdd(SFP + 48, (unsigned long long)(clock64() - SFX0));\n    if (tid >= 128) {\n      const int ht = tid - 128;"),
    (PFL, "    __syncthreads();\n    if (!ovl) {\n      outside_tables(tid, SF_PFL_NT);",
     "    if (SFL) atomicAdd(SFP + 49, (unsigned long long)(clock64() - SFX0));\n    __syncthreads();\n"
     "    if (SFL) { atomicAdd(SFP + 50, (unsigned long long)(clock64() - SFX0)); atomicAdd(SFP + 51, 1ull); }\n    if (!ovl) {\n      outside_tables(tid, SF_PFL_NT);"),
    (PFL, "      if (team == 0) {\n        if (valid) {\n          // (the cell's weights first:",
     "      if (SFL) atomicAdd(SFB + 0, (unsigned long long)(clock64() - SFT0));\n"
     "      if (team == 0) {\n        if (valid) {\n          // (the cell's weights first:"),
    (PFL, "    const bool nbL = SH && pos > 0, nbR = SH && pos + W < L;\n    __syncthreads();\n",
     "    const bool nbL = SH && pos > 0, nbR = SH && pos + W < L;\n    __syncthreads();\n"
     "    unsigned long long *const SFP = (unsigned long long *)((char *)status + 32768) + (tid >> 6) * 64;\n"
     "    const bool SFL = status && (tid & 63) == 0 && (blockIdx.x & 63) == 5;\n"
     "    const bool SFL0 = SFL && tid == 0;\n"
     "    long long SFF = 0;\n"
     "#define SF_PH(slot) if (SFL0) { const long long t_ = (long long)clock64(); atomicAdd(SFP + (slot), (unsigned long long)(t_ - SFF)); SFF = t_; }\n"),
    (PFL, "    double H[27];\n#pragma unroll\n    for (int u = 0; u < 27; u++) H[u] = 0.0;\n    if (resume) {",
     "    double H[27];\n#pragma unroll\n    for (int u = 0; u < 27; u++) H[u] = 0.0;\n    SFF = SFL ? (long long)clock64() : 0;\n    if (resume) {"),
    (PFL, "    for (int j = resume ? W - step + 1 : SFD_TURN + 2; j <= W + 1; j++) {\n",
     "    SF_PH(40)\n    for (int j = resume ? W - step + 1 : SFD_TURN + 2; j <= W + 1; j++) {\n      const long long SFT0 = SFL ? (long long)clock64() : 0;\n"),
    (PFL, "      __syncthreads();\n      if (team == 2 && qvalid) QMD(dq, i) = ZP[5 * VW + i] + ZP[4 * VW + i];",
     "      if (SFL) atomicAdd(SFP + 32, (unsigned long long)(clock64() - SFT0));\n      __syncthreads();\n"
     "      if (SFL) atomicAdd(SFP + 33, (unsigned long long)(clock64() - SFT0));\n"
     "      if (team == 2 && qvalid) QMD(dq, i) = ZP[5 * VW + i] + ZP[4 * VW + i];"),
    (PFL, "        qm1c[i] = m1;\n      }\n      __syncthreads();\n    }\n",
     "        qm1c[i] = m1;\n      }\n      if (SFL) atomicAdd(SFP + 34, (unsigned long long)(clock64() - SFT0));\n      __syncthreads();\n"
     "      if (SFL) { atomicAdd(SFP + 35, (unsigned long long)(clock64() - SFT0)); atomicAdd(SFP + 36, 1ull); }\n    }\n    SF_PH(41)\n"),
    (PFL, "    for (int l = W; l >= SFD_TURN + 2; l--) {\n",
     "    SF_PH(42)\n    for (int l = W; l >= SFD_TURN + 2; l--) {\n      const long long SFT0 = SFL ? (long long)clock64() : 0;\n"
     "      unsigned long long *const SFB = SFP + ((l - 1) / 30) * 8;\n"),
    (PFL, "      __syncthreads();\n      SF_LANE_TABLE_LOAD(tpk, L, BWD[sfd_min(l - 1 + L, W)]);",
     "      if (SFL) atomicAdd(SFB + 1, (unsigned long long)(clock64() - SFT0));\n      __syncthreads();\n"
     "      if (SFL) atomicAdd(SFB + 2, (unsigned long long)(clock64() - SFT0));\n      SF_LANE_TABLE_LOAD(tpk, L, BWD[sfd_min(l - 1 + L, W)]);"),
    (PFL, "          } else cd += p;\n        }\n      }\n      SF_LANE_TABLE_PIN(tpk);\n      SF_LANE_TABLE_PIN(tcol);\n      __syncthreads();\n    }\n",
     "          } else cd += p;\n        }\n      }\n      SF_LANE_TABLE_PIN(tpk);\n      SF_LANE_TABLE_PIN(tcol);\n      if (SFL) atomicAdd(SFB + 3, (unsigned long long)(clock64() - SFT0));\n      __syncthreads();\n"
     "      if (SFL) { atomicAdd(SFB + 4, (unsigned long long)(clock64() - SFT0)); atomicAdd(SFB + 5, 1ull); }\n    }\n    SF_PH(43)\n"),
    (PFL, "      if (centroid_dist) centroid_dist[fold] = cd;\n    }\n",
     "      if (centroid_dist) centroid_dist[fold] = cd;\n    }\n    SF_PH(44)\n    if (SFL0) atomicAdd(SFP + 45, 1ull);\n"),
    (HOST, "    int rc = ensure(g.status, sizeof(int));\n    if (rc) return rc;\n    HIPCHK(hipMemset(g.status.p, 0, sizeof(int)));",
     "    int rc = ensure(g.status, 65536);\n    if (rc) return rc;\n    HIPCHK(hipMemset(g.status.p, 0, 65536));"),
    (HOST, "                     d_dG, d_mbd, d_cen, d_cd, d_tr, L, win0, step, run_len, share, (const char *)nullptr, (int *)nullptr);\n",
     "                     d_dG, d_mbd, d_cen, d_cd, d_tr, L, win0, step, run_len, share, (const char *)nullptr, (int *)g.status.p);\n"
     "    if (const char *so = getenv(\"SF_STAMP_OUT\")) {\n      static unsigned long long hb[8 * 64];\n      HIPCHK(hipStreamSynchronize(st));\n"
     "      HIPCHK(hipMemcpy(hb, (char *)g.status.p + 32768, sizeof hb, hipMemcpyDeviceToHost));\n"
     "      HIPCHK(hipMemset((char *)g.status.p + 32768, 0, sizeof hb));\n      if (FILE *f = fopen(so, \"a\")) {\n"
     "        fprintf(f, \"launch n %d W %d run_len %d\\n\", n, W, run_len);\n"
     "        for (int k = 0; k < 8; k++) { for (int q = 0; q < 64; q++) fprintf(f, \"%llu \", hb[k * 64 + q]); fprintf(f, \"\\n\"); }\n"
     "        fclose(f);\n      }\n    }\n"),
]

VARIANTS = {
    "pfstamps": (PFSTAMP_PATCHES, []),
    # partition function, outside pass: team 1's share of team 3's multiloop sum, blocks of eight terms: (l - MLS0) / MLS1
    # partition function, outside pass: the columns in which team 1 takes six of team 0's special loops
    "pflsp0": ([], ["-DSF_PFL_LSP=0"]), "pflsp40": ([], ["-DSF_PFL_LSP=40"]), "pflsp90": ([], ["-DSF_PFL_LSP=90"]),
    "pfmls30_8": ([], ["-DSF_PFL_MLS0=30", "-DSF_PFL_MLS1=8"]),
    "pfmls20_10": ([], ["-DSF_PFL_MLS0=20", "-DSF_PFL_MLS1=10"]),
    "pfmls10_12": ([], ["-DSF_PFL_MLS0=10", "-DSF_PFL_MLS1=12"]),
    "pfmls40_8": ([], ["-DSF_PFL_MLS0=40", "-DSF_PFL_MLS1=8"]),
    "stamps": (STAMP_PATCHES, []),
    # name: (patches, flags)
    "head": ([], []),
    "nb6": ([(FAST, "#define SF_HELP_NB_128 4", "#define SF_HELP_NB_128 6")], []),
    "nb8": ([(FAST, "#define SF_HELP_NB_128 4", "#define SF_HELP_NB_128 8")], []),
    "nb2": ([(FAST, "#define SF_HELP_NB_128 4", "#define SF_HELP_NB_128 2")], []),
    "gather16": ([(FAST, "  (FOLD ? sf_gather16_at((F), (unsigned)offsetof(SfFastParams, field) + ((unsigned)(idx) << 1)) : (int)(F)->field[idx])", "  (sf_gather16_at((F), (unsigned)offsetof(SfFastParams, field) + ((unsigned)(idx) << 1)))")], []),
    "ntload": ([(FAST, "    for (int x = tidf; x < W; x += NT) S[x + 1] = sf_encode_nt(src[x]);", "    for (int x = tidf; x < W; x += NT) S[x + 1] = sf_encode_nt(__builtin_nontemporal_load(&src[x]));")], []),
    # (the scratch stride: "cgNNNN": ([(FAST, "#define SF_CG_ENTRIES(W) ((W) == 120 ? 7168 :", "#define SF_CG_ENTRIES(W) ((W) == 120 ? NNNN :")], []) —
    #  round 5 re-swept 6796 .. 8192 with tools/pmc_cmp.sh "WRITE_SIZE": 7168 still writes the least, 0.44 kB per fold against 0.78 .. 3.8)
    "tiny8": ([], ["-DSF_FAST_TINY_D0=8"]), "tiny12": ([], ["-DSF_FAST_TINY_D0=12"]),
    "pb4": ([], ["-DSF_UNP_PB=4"]),
    "pb6": ([], ["-DSF_UNP_PB=6"]), "pb12": ([], ["-DSF_UNP_PB=12"]),
    "nounpack": ([], ["-DSF_FAST_UNPACK=0"]),
    # sections: instructions per fold of a section = product - ablation, by SQ counters (tools/pmc_cmp.sh; profiles/r05/mfe_section_budget.txt)
    "secP1": ([(FAST, "  if (SEC & SF_SEC_P1) {\n", "  if (false) {\n")], []),
    "secHELP": ([(FAST, "  if (SEC & SF_SEC_HELP) {\n", "  eh = SF_FAST_BIG;\n  if (false) {\n")], []),
    "secBN": ([(FAST, "      int gb = SF_FAST_BIG, g1 = SF_FAST_BIG;\n      if (G) {", "      int gb = SF_FAST_BIG, g1 = SF_FAST_BIG;\n      if (true) {\n      } else if (G) {")], []),
    "secDML1": ([(FAST, "    dec = SF_FAST_BIG;\n  if (FOLD) {", "    dec = SF_FAST_BIG;\n  if (true) return;\n  if (FOLD) {")], []),
    "secC0": ([(FAST, "  if (SEC & SF_SEC_C0) {\n", "  if (SEC & SF_SEC_C0) e0 = SF_FAST_BIG;\n  if (false) {\n")], []),
    "secSWEEP": ([(FAST, "      if (sweep_now && sweep_rows > 0) sf_trail_rows<NQ, DROWS>(T, tid & 63, sweep_rows, dc);", "      (void)0;"),
                  (FAST, "      const bool sweep_now = defer_on && sweeper && split && T.row > 0;", "      const bool sweep_now = false;")], []),
    # phases: which steps cost what (the other phase keeps its barriers and control code)
    "nosplitwork": ([(FAST, CELLS, "      if (!split && __ballot(valid)) {\n        if (DO_G && d0 < 8)"),
                     (FAST, FIN, "        if (false) {")], []),
    "nounsplitwork": ([(FAST, CELLS, "      if (split && __ballot(valid)) {\n        if (DO_G && d0 < 8)")], []),
    # split steps, by role
    "nodml2": ([(FAST, DML2, "            dec = SF_INF16;")], []),
    "nomainp1": ([(FAST, MAINCALL, MAINCALL.replace("SF_SEC_P1 | SF_SEC_C0 | SF_SEC_PRE", "SF_SEC_PRE"))], []),
    "nomgh": ([(FAST, MGHCALL, "            if (false) " + MGHCALL.strip())], []),
    "nofin": ([(FAST, FIN, "        if (false) {")], []),
    # what the main waves' post-barrier work costs, piece by piece
    "nofixupS": ([(FAST, "      if (grp == 0 && valid && !helper && !dmlw) {", "      if (!split && grp == 0 && valid && !helper && !dmlw) {")], []),
    "noscratchS": ([(FAST, "  X.cg[SF_CGIDX(i, j)] = (int16_t)cx;", "  if (!(SEC & SF_SEC_POST)) X.cg[SF_CGIDX(i, j)] = (int16_t)cx;")], []),
    # the phases before the split steps
    "noG": ([(FAST, CELLS, "      if (d0 >= SF_FAST_TINY_D0 && __ballot(valid)) {\n        if (DO_G && d0 < 8)")], []),
    "noCH": ([(FAST, CELLS, "      if ((d0 < SF_FAST_TINY_D0 || d0 >= SF_FAST_CHUNK_D0) && __ballot(valid)) {\n        if (DO_G && d0 < 8)")], []),
    "noU36": ([(FAST, CELLS, "      if ((d0 < SF_FAST_CHUNK_D0 || split) && __ballot(valid)) {\n        if (DO_G && d0 < 8)")], []),
}


def build(name):
    patches, flags = VARIANTS[name]
    tmp = "/tmp/abl_" + name
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp + "/scanfold_amd")
    shutil.copytree(os.path.join(ROOT, "scanfold_amd", "csrc"), tmp + "/scanfold_amd/csrc", ignore=shutil.ignore_patterns("*.so"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp + "/include")
    for rel, anchor, repl in patches:
        path = os.path.join(tmp, rel)
        s = open(path).read()
        if s.count(anchor) != 1:
            return name, "anchor occurs %d times: %r" % (s.count(anchor), anchor[:60])
        open(path, "w").write(s.replace(anchor, repl))
    out = os.path.join(ROOT, "tools", "abl_%s.so" % name)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"] + flags + \
          ["-o", out, tmp + "/scanfold_amd/csrc/scanfold_hip.hip"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    return name, "ok" if p.returncode == 0 else p.stderr[-1500:]


if __name__ == "__main__":
    names = sys.argv[1:] or sorted(VARIANTS)
    with ThreadPoolExecutor(max_workers=6) as ex:
        for name, msg in ex.map(build, names):
            print(name, msg)
