"""Generates the EXPERIMENT sources of the edge-loss investigation from the
product sources: humid_amd/csrc -> tools/exp_plan/csrc (not tracked).  k_combo_keys and k_pairs get
the pigeonhole plan a second and third time -- as a ~0.7 KB struct BY VALUE (`xv`) and as a pointer
to a copy uploaded with hipMemcpyAsync (`xp`) -- and compile-time switches choose, per use, where
the kernel reads from:

  -DEXP_KEYS=f   fields of k_combo_keys          f = 0 product form (small by-value struct, static
  -DEXP_MASK=f   bucket mask of k_pairs              index), 1 xv (dynamic index into the kernarg
  -DEXP_EM=f     earlier-combination masks           segment), 2 xp (uploaded copy, uniform loads)
  -DEXP_VMEM=1   forms 1/2 load through vector memory (a lane-dependent zero defeats the
                 uniformity analysis, so no scalar load / scalar cache is involved)
  -DEXP_NOBREAK=1  the earlier-masks loop runs to the end instead of leaving at the first hit
  env EXP_SYNC_UPLOAD=1 (run time): form 2's upload is hipMemcpy + hipDeviceSynchronize

    python tools/exp_plan/make_exp.py            # sources
    python tools/exp_plan/make_exp.py build      # + every variant in VARIANTS -> libexp_<name>.so
"""
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "humid_amd", "csrc")
DST = os.path.join(HERE, "csrc")

VARIANTS = {
    "ctl": "",                                      # all product forms, extra arguments passed along
    "k1": "-DEXP_KEYS=1",
    "m1": "-DEXP_MASK=1",
    "e1": "-DEXP_EM=1",
    "e2": "-DEXP_EM=2",
    "e1_vmem": "-DEXP_EM=1 -DEXP_VMEM=1",
    "e2_vmem": "-DEXP_EM=2 -DEXP_VMEM=1",
    "e1_nobreak": "-DEXP_EM=1 -DEXP_NOBREAK=1",
    "all1": "-DEXP_KEYS=1 -DEXP_MASK=1 -DEXP_EM=1",
    "all2": "-DEXP_KEYS=2 -DEXP_MASK=2 -DEXP_EM=2",
}


def sub1(s, old, new):
    assert s.count(old) == 1, (s.count(old), old[:60])
    return s.replace(old, new)


def gen():
    if os.path.exists(DST):
        shutil.rmtree(DST)
    shutil.copytree(SRC, DST, ignore=shutil.ignore_patterns("host"))
    p = os.path.join(DST, "common.hip.h")
    s = open(p).read()
    s = sub1(s, '#include "../../include/humid_hip.h"', '#include "../../../include/humid_hip.h"')
    open(p, "w").write(s)

    p = os.path.join(DST, "kernels_graph.hip.h")
    s = open(p).read()
    s = sub1(s, "template <class KeyT, class WT>\n__global__ void k_combo_keys(", '''#ifndef EXP_KEYS
#define EXP_KEYS 0
#endif
#ifndef EXP_MASK
#define EXP_MASK 0
#endif
#ifndef EXP_EM
#define EXP_EM 0
#endif
#ifndef EXP_VMEM
#define EXP_VMEM 0
#endif
#ifndef EXP_NOBREAK
#define EXP_NOBREAK 0
#endif
// form f of plan member `field`: 1 = by-value copy in the kernarg segment, 2 = uploaded copy
#define EXP_GET(f, field) ((f) == 1 ? xv.field : xp->field)
#if EXP_VMEM
#define EXP_Z (__builtin_amdgcn_mbcnt_lo(0u, 0u))        // 0 in every lane, divergent to the compiler
#else
#define EXP_Z 0u
#endif
template <class KeyT, class WT>
__global__ void k_combo_keys(''')
    s = sub1(s, "k_combo_keys(const WT *__restrict__ s_word, u32 n, ComboFields cf,\n",
             "k_combo_keys(const WT *__restrict__ s_word, u32 n, ComboFields cf, ComboPlan xv, const ComboPlan *__restrict__ xp, u32 xcb,\n")
    s = sub1(s, '''  const WT w = s_word[i];
  u64 k = 0;
#pragma unroll
  for (u32 f = 0; f < MAX_FIELDS; f++) {
    if (f < cf.nf) {
      const u32 wd = cf.width[f];
      k = ((wd >= 64) ? 0ull : (k << wd)) | w_field(w, cf.shift[f], wd);
    }
  }
  key[i] = (KeyT)k;
  val[i] = i;''', '''  const WT w = s_word[i];
  u64 k = 0;
#if EXP_KEYS == 0
  (void)xv; (void)xp; (void)xcb;
#pragma unroll
  for (u32 f = 0; f < MAX_FIELDS; f++) {
    if (f < cf.nf) {
      const u32 wd = cf.width[f];
      k = ((wd >= 64) ? 0ull : (k << wd)) | w_field(w, cf.shift[f], wd);
    }
  }
#else
  const u32 nf = EXP_GET(EXP_KEYS, nfield[xcb + EXP_Z]);
  for (u32 f = 0; f < nf; f++) {
    const u32 wd = EXP_GET(EXP_KEYS, width[xcb + EXP_Z][f]);
    k = ((wd >= 64) ? 0ull : (k << wd)) | w_field(w, EXP_GET(EXP_KEYS, shift[xcb + EXP_Z][f]), wd);
  }
#endif
  key[i] = (KeyT)k;
  val[i] = i;''')
    s = sub1(s, "        EarlierMasksT<WT> em, u32 cb, u32 distance, u32 *deg, u32 *parent,\n",
             "        EarlierMasksT<WT> em, ComboPlan xv, const ComboPlan *__restrict__ xp, u32 cb, u32 distance, u32 *deg, u32 *parent,\n")
    s = sub1(s, "k_pairs(const WT *__restrict__ W, const u32 *__restrict__ V, u32 n, u32 i0, u32 n_i, WT mask,\n",
             "k_pairs(const WT *__restrict__ W, const u32 *__restrict__ V, u32 n, u32 i0, u32 n_i, WT mask_arg,\n")
    s = sub1(s, '''  const u32 ri = PASS0 ? i : V[i];
  u32 found = 0;''', '''  const u32 ri = PASS0 ? i : V[i];
#if EXP_MASK == 0
  const WT mask = mask_arg;
#else
  const WT mask = w_from<WT>(EXP_GET(EXP_MASK, mask[cb + EXP_Z]));
#endif
  u32 found = 0;''')
    s = sub1(s, '''    bool first = true;
#pragma unroll
    for (u32 q = 0; q < MAX_COMBOS; q++)
      first = first && !(q < cb && !w_hits(x, em.m[q]));
    if (!first) continue;''', '''    bool first = true;
#if EXP_EM == 0
#pragma unroll
    for (u32 q = 0; q < MAX_COMBOS; q++)
      first = first && !(q < cb && !w_hits(x, em.m[q]));
#else
    for (u32 q = 0; q < cb; q++)
      if (!w_hits(x, w_from<WT>(EXP_GET(EXP_EM, mask[q + EXP_Z])))) {
        first = false;
#if !EXP_NOBREAK
        break;
#endif
      }
#endif
    if (!first) continue;''')
    open(p, "w").write(s)

    p = os.path.join(DST, "humid_hip.hip")
    s = open(p).read()
    s = sub1(s, "seg_ws, csize, cur;\n", "seg_ws, csize, cur, plan_dev;\n  ComboPlan h_plan;          // form 2: the host copy the asynchronous upload reads\n")
    s = sub1(s, "// ---- cluster stage shared by the full pipeline", '''// experiment build: the plan travels a second time by value and a third time as an uploaded copy.
// stage_graph uploads asynchronously from a member of the context (the round-1 form); the other
// callers, not exercised by the experiment, upload synchronously.
static const ComboPlan *exp_upload(humid_ctx *c, const ComboPlan &plan, bool async) {
  if (c->plan_dev.ensure(sizeof(ComboPlan)) != hipSuccess) return nullptr;
  if (async && !getenv("EXP_SYNC_UPLOAD")) {
    c->h_plan = plan;
    (void)hipMemcpyAsync(c->plan_dev.p, &c->h_plan, sizeof(ComboPlan), hipMemcpyHostToDevice, c->stream);
  } else {
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpy(c->plan_dev.p, &plan, sizeof(ComboPlan), hipMemcpyHostToDevice);
    (void)hipDeviceSynchronize();
  }
  return c->plan_dev.as<ComboPlan>();
}

// ---- cluster stage shared by the full pipeline''')
    # stage_graph: one asynchronous upload per call, then every launch gets (plan, pointer)
    s = sub1(s, "  EarlierMasksT<WT> d_masks;                         // masks of all combos, for the first-combo rule\n",
             "  const ComboPlan *XP = exp_upload(c, plan, true);\n  EarlierMasksT<WT> d_masks;                         // masks of all combos, for the first-combo rule\n")
    i0 = s.index("static int stage_graph(")
    i1 = s.index("static ComboFields plan_fields(const ComboPlan &plan, u32 cb);")
    g = s[i0:i1]
    g = g.replace("d_masks, seg, distance", "d_masks, plan, XP, seg, distance")
    g = g.replace("g_word, U, fields_of(seg),", "g_word, U, fields_of(seg), plan, XP, seg,")
    s = s[:i0] + g + s[i1:]
    # the remaining callers
    i2 = s.index("static ComboFields plan_fields(const ComboPlan &plan, u32 cb);")
    head, rest = s[:i2], s[i2:]
    rest = re.sub(r"d_masks,(\s)", r"d_masks, plan, exp_upload(c, plan, false),\1", rest)
    rest = rest.replace("g_word, U, cf,\n", "g_word, U, cf, plan, exp_upload(c, plan, false), 0u,\n")
    rest = rest.replace("plan_fields(plan, combo), c->seg_k0", "plan_fields(plan, combo), plan, exp_upload(c, plan, false), combo, c->seg_k0")
    s = head + rest
    open(p, "w").write(s)


def build(names):
    for name in names:
        out = os.path.join(HERE, "libexp_%s.so" % name)
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + VARIANTS[name].split() + \
              ["-o", out, os.path.join(DST, "humid_hip.hip")]
        print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)


if __name__ == "__main__":
    gen()
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build(sys.argv[2:] or list(VARIANTS))
