#!/usr/bin/env python3
"""oracle/apply_dispatch.py -- TEST INFRASTRUCTURE.

Applies the insertions of INTEGRATION.md section 2 to COPIES of seven reference source files (the copies live under
oracle/_ref/patched/, git-ignored; nothing under /root/reference is touched and no reference text is kept in this
repository: every insertion is located by a one-line anchor).  Insertions only -- no reference line is changed or
removed -- and all of them sit under  #if defined (CHOLMOD_HAS_MI355) && defined (CHOLMOD_INT64).

usage: apply_dispatch.py <reference CHOLMOD dir> <output dir>
"""
import os
import re
import sys

GUARD = "#if defined ( CHOLMOD_HAS_MI355 ) && defined ( CHOLMOD_INT64 )"
INCLUDE = ["#ifdef CHOLMOD_HAS_MI355", '#include "cholmod_mi355.h"', "#endif"]

# file -> list of (anchor regex, which occurrence (0-based), lines inserted BEFORE the anchor line)
EDITS = {
    "Supernodal/cholmod_super_symbolic.c": [
        (r'^#include "cholmod_internal\.h"', 0, None),            # None: INCLUDE goes AFTER this line
        # the branch a build without CUDA takes ("GPU module is not installed": Common->useGPU = 0)
        (r"^\s*#else\s*$", "before:GPU module is not installed", [
            "    #elif defined ( CHOLMOD_HAS_MI355 ) && defined ( CHOLMOD_INT64 )",
            "    cholmod_mi355_select (A, for_whom, L, Common) ;",
        ]),
    ],
    "Supernodal/cholmod_super_numeric.c": [
        (r'^#include "cholmod_internal\.h"', 0, None),
        (r"^\s*float s_beta \[2\] ;", 0, [
            "    " + GUARD,
            "    const int mi355_took = cholmod_mi355_super_numeric (A, F, beta, L, Common, &ok) ;",
            "    #else",
            "    const int mi355_took = FALSE ;",
            "    #endif",
        ]),
        (r"^\s*switch \(\(A->xtype \+ A->dtype\) % 8\)", 0, [
            "    if (!mi355_took)     /* else: the CPU workers below are skipped */",
        ]),
    ],
    "Supernodal/cholmod_super_solve.c": [
        (r'^#include "cholmod_internal\.h"', 0, None),
        (r"^\s*switch \(\(L->xtype \+ L->dtype\) % 8\)", 0, [
            "    " + GUARD,
            "    { int mi355_ok ; if (cholmod_mi355_super_solve (L, X, 1, Common, &mi355_ok)) return (mi355_ok) ; }",
            "    #endif",
        ]),
        (r"^\s*switch \(\(L->xtype \+ L->dtype\) % 8\)", 1, [
            "    " + GUARD,
            "    { int mi355_ok ; if (cholmod_mi355_super_solve (L, X, 0, Common, &mi355_ok)) return (mi355_ok) ; }",
            "    #endif",
        ]),
    ],
    "Cholesky/cholmod_factorize.c": [
        (r'^#include "cholmod_internal\.h"', 0, None),
        (r"^\s*if \(L->ordering == CHOLMOD_NATURAL\)", 0, [    # supernodal branch of cholmod_factorize_p
            "        " + GUARD,
            "        if (cholmod_mi355_defer_permutation (A, L, Common))",
            "        {",
            "            S = A ;     /* the device places A through L->Perm: no permuted copy is built */",
            "        }",
            "        else",
            "        #endif",
        ]),
        (r"^\s*status = Common->status ;", 0, [                 # right after the cholmod_super_numeric call
            "        " + GUARD,
            "        cholmod_mi355_defer_clear ( ) ;",
            "        #endif",
        ]),
    ],
    "Cholesky/cholmod_solve.c": [
        (r'^#include "cholmod_internal\.h"', 0, None),
        (r"^\s*cholmod_dense \*E ;", 0, [                  # head of the supernodal branch of cholmod_solve2
            "        " + GUARD,
            "        { int mi355_ok ; if (cholmod_mi355_solve2 (sys, L, B, X, Common, &mi355_ok)) return (mi355_ok) ; }",
            "        #endif",
        ]),
    ],
    "Utility/t_cholmod_free_factor.c": [
        (r'^#include "cholmod_internal\.h"', 0, None),
        (r"^\s*size_t ei = sizeof \(Int\) ;", 0, [         # first statement of cholmod_to_simplicial_sym
            "    " + GUARD,
            "    cholmod_mi355_forget (L) ;",
            "    #endif",
        ]),
    ],
    "Utility/t_cholmod_change_factor.c": [
        (r'^#include "cholmod_internal\.h"', 0, None),
        (r"^\s*L->is_super = FALSE ;\s*// L is now simplicial", 0, [     # super_num_to_simplicial_num
            "    " + GUARD,
            "    cholmod_mi355_forget (L) ;",
            "    #endif",
        ]),
    ],
}


def patch(text, edits, name):
    lines = text.split("\n")
    for anchor, which, ins in edits:
        rx = re.compile(anchor)
        hits = [i for i, ln in enumerate(lines) if rx.search(ln)]
        if isinstance(which, str):                      # "before:<text on the following line>"
            want = which.split(":", 1)[1]
            hits = [i for i in hits if i + 1 < len(lines) and want in lines[i + 1]]
            which = 0
        if len(hits) <= which:
            raise SystemExit(f"apply_dispatch: anchor {anchor!r} (occurrence {which}) not found in {name}")
        at = hits[which]
        if ins is None:
            lines[at + 1:at + 1] = INCLUDE
        else:
            lines[at:at] = ins
    return "\n".join(lines)


def main():
    ref, out = sys.argv[1], sys.argv[2]
    for rel, edits in EDITS.items():
        src = open(os.path.join(ref, rel)).read()
        dst = os.path.join(out, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        new = patch(src, edits, rel)
        if not os.path.exists(dst) or open(dst).read() != new:
            open(dst, "w").write(new)
        print(f"apply_dispatch: {rel}: +{new.count(chr(10)) - src.count(chr(10))} lines")


if __name__ == "__main__":
    main()
