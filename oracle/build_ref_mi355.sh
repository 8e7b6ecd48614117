#!/usr/bin/env bash
# oracle/build_ref_mi355.sh -- TEST INFRASTRUCTURE, not product code.
#
# The reference CHOLMOD with the drop-in wired in, exactly as INTEGRATION.md describes it:
#   * oracle/apply_dispatch.py inserts the hooks into COPIES of seven reference sources (under
#     oracle/_ref/patched/, git-ignored; /root/reference is read only);
#   * those copies are compiled with -DCHOLMOD_HAS_MI355 (and -DBLAS_TIMER, so that the Common->cholmod_gpu_*
#     times are filled) together with OUR glue suitesparse_amd/csrc/cholmod_mi355.c;
#   * every other object is the unchanged reference's, reused from oracle/_ref/obj (build_ref.sh runs first).
# Output: oracle/_ref/libcholmod_mi355.so + oracle/_ref/libref_helper_mi355.so (the same flat helper, linked to it).
# The mi355_* symbols stay UNDEFINED in the library: whoever loads it loads the engine first with RTLD_GLOBAL --
# tests/helpers/libmi355_cpuemu.so on a machine without a GPU, suitesparse_amd/libmi355_cholmod.so on the GPU box.
set -euo pipefail
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
OUT=$HERE/_ref
OBJ=$OUT/obj
PATCHED=$OUT/patched
POBJ=$OUT/obj_mi355
BLASLIB=${BLASLIB:-/opt/conda/lib/libmkl_rt.so}

if [ ! -d "$REF/CHOLMOD" ]; then
    echo "build_ref_mi355: $REF/CHOLMOD not present (GPU box?) -- using prebuilt files" >&2
    exit 0
fi
[ -f "$OUT/libcholmod_ref.so" ] || bash "$HERE/build_ref.sh"
mkdir -p "$PATCHED" "$POBJ"
python3 "$HERE/apply_dispatch.py" "$REF/CHOLMOD" "$PATCHED"

# include path: the patched copies' directories come first ONLY for the files they replace (the l_ wrappers include
# them by name), then the reference's
INC="-I$ROOT/include -I$PATCHED/Supernodal -I$PATCHED/Utility -I$PATCHED/Cholesky \
 -I$REF/SuiteSparse_config -I$REF/AMD/Include -I$REF/CAMD/Include \
 -I$REF/COLAMD/Include -I$REF/CCOLAMD/Include -I$REF/CHOLMOD/Include -I$REF/CHOLMOD \
 -I$REF/CHOLMOD/Check -I$REF/CHOLMOD/Cholesky -I$REF/CHOLMOD/Utility \
 -I$REF/CHOLMOD/MatrixOps -I$REF/CHOLMOD/Modify -I$REF/CHOLMOD/Partition \
 -I$REF/CHOLMOD/Supernodal -I$REF/CHOLMOD/SuiteSparse_metis/GKlib \
 -I$REF/CHOLMOD/SuiteSparse_metis/libmetis -I$REF/CHOLMOD/SuiteSparse_metis/include"
CFLAGS="-O2 -fPIC -fopenmp -std=c11 -w -DBLAS32 -D_GNU_SOURCE -D_DEFAULT_SOURCE -DCHOLMOD_HAS_MI355 -DBLAS_TIMER"

# the int64 AND int32 instantiations of the seven files, compiled the way the reference's three-line wrappers
# (cholmod_l_xxx.c: "#define CHOLMOD_INT64" + include) instantiate them -- the define goes on the command line, because
# a quoted include from a wrapper under $REF would find the UNPATCHED file next to it first.  The int32 objects
# compile the hooks away (CHOLMOD_INT64 is not defined there); they are rebuilt so that no object of the unchanged
# build is mixed with a patched twin.
REPLACED=""
build_pair() {   # $1 dir, $2 patched source, $3 object stem, $4 int32 define ("" = none)
    gcc $CFLAGS $4 $INC -c "$PATCHED/$1/$2" -o "$POBJ/CHOLMOD_$1_cholmod_$3.o"
    gcc $CFLAGS -DCHOLMOD_INT64 $INC -c "$PATCHED/$1/$2" -o "$POBJ/CHOLMOD_$1_cholmod_l_$3.o"
    REPLACED="$REPLACED CHOLMOD_$1_cholmod_$3.o CHOLMOD_$1_cholmod_l_$3.o"
}
build_pair Supernodal cholmod_super_symbolic.c super_symbolic ""
build_pair Supernodal cholmod_super_numeric.c  super_numeric  ""
build_pair Supernodal cholmod_super_solve.c    super_solve    ""
build_pair Cholesky   cholmod_solve.c          solve          ""
build_pair Cholesky   cholmod_factorize.c      factorize      ""
build_pair Utility    t_cholmod_free_factor.c   free_factor   -DCHOLMOD_INT32
build_pair Utility    t_cholmod_change_factor.c change_factor -DCHOLMOD_INT32
gcc $CFLAGS -Wall $INC -c "$ROOT/suitesparse_amd/csrc/cholmod_mi355.c" -o "$POBJ/cholmod_mi355.o"

KEEP=""
for o in "$OBJ"/*.o; do
    b=$(basename "$o")
    case " $REPLACED " in *" $b "*) ;; *) KEEP="$KEEP $o" ;; esac
done
BLASDIR=$(dirname "$BLASLIB")
gcc -shared -fopenmp -Wl,-Bsymbolic -o "$OUT/libcholmod_mi355.so" $KEEP "$POBJ"/*.o \
    "$BLASLIB" -Wl,-rpath,"$BLASDIR" -lm -ldl -lpthread
gcc -O2 -fPIC -fopenmp -std=c11 -Wall $INC -DBLAS32 -DREF_HELPER_MI355 -shared \
    -o "$OUT/libref_helper_mi355.so" "$HERE/ref_helper.c" \
    -L"$OUT" -lcholmod_mi355 -Wl,-rpath,'$ORIGIN' -Wl,-rpath,"$BLASDIR" -lm
echo "build_ref_mi355: OK -> $OUT/libcholmod_mi355.so $OUT/libref_helper_mi355.so"
