#!/usr/bin/env bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE, not product code.
#
# Compiles the UNCHANGED reference CHOLMOD (plus SuiteSparse_config, AMD, CAMD,
# COLAMD, CCOLAMD and the METIS copy CHOLMOD embeds) from the sources where they
# lie under $REF (default /root/reference) with plain gcc -- the reference's own
# build system (cmake) is NOT run and nothing is written under $REF.  Outputs go
# only to oracle/_ref/ (git-ignored, but shipped to the GPU box by gpurun):
#
#   oracle/_ref/libcholmod_ref.so   reference CHOLMOD 5.3.0, CPU path, host BLAS
#   oracle/_ref/libref_helper.so    flat-array C helper around it (ref_helper.c)
#
# Source list = the globs of CHOLMOD/CMakeLists.txt:372-374; include dirs =
# CHOLMOD/CMakeLists.txt:365-366,569-571.  BLAS/LAPACK: the image's MKL
# (/opt/conda/lib/libmkl_rt.so, LP64 => -DBLAS32), as in SURVEY.md Appendix A.
set -euo pipefail
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
OBJ=$OUT/obj
BLASLIB=${BLASLIB:-/opt/conda/lib/libmkl_rt.so}
JOBS=${JOBS:-8}

if [ ! -d "$REF/CHOLMOD" ]; then
    echo "build_ref: $REF/CHOLMOD not present (GPU box?) -- using prebuilt files" >&2
    exit 0
fi
mkdir -p "$OBJ"

INC="-I$REF/SuiteSparse_config -I$REF/AMD/Include -I$REF/CAMD/Include \
 -I$REF/COLAMD/Include -I$REF/CCOLAMD/Include -I$REF/CHOLMOD/Include -I$REF/CHOLMOD \
 -I$REF/CHOLMOD/Check -I$REF/CHOLMOD/Cholesky -I$REF/CHOLMOD/Utility \
 -I$REF/CHOLMOD/MatrixOps -I$REF/CHOLMOD/Modify -I$REF/CHOLMOD/Partition \
 -I$REF/CHOLMOD/Supernodal -I$REF/CHOLMOD/SuiteSparse_metis/GKlib \
 -I$REF/CHOLMOD/SuiteSparse_metis/libmetis -I$REF/CHOLMOD/SuiteSparse_metis/include"
CFLAGS="-O2 -fPIC -fopenmp -std=c11 -w -DBLAS32 -D_GNU_SOURCE -D_DEFAULT_SOURCE"

SRCS=$(ls "$REF"/SuiteSparse_config/SuiteSparse_config.c \
          "$REF"/AMD/Source/*.c "$REF"/CAMD/Source/*.c \
          "$REF"/COLAMD/Source/*.c "$REF"/CCOLAMD/Source/*.c \
          "$REF"/CHOLMOD/Check/cholmod_*.c "$REF"/CHOLMOD/Cholesky/cholmod_*.c \
          "$REF"/CHOLMOD/Utility/cholmod_*.c "$REF"/CHOLMOD/MatrixOps/cholmod_*.c \
          "$REF"/CHOLMOD/Modify/cholmod_*.c "$REF"/CHOLMOD/Partition/cholmod_*.c \
          "$REF"/CHOLMOD/Supernodal/cholmod_*.c)

compile_one() {
    src=$1
    rel=${src#"$REF"/}
    obj="$OBJ/$(echo "$rel" | tr '/' '_' | sed 's/\.c$/.o/')"
    if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ]; then
        gcc $CFLAGS $INC -c "$src" -o "$obj"
    fi
}
export -f compile_one
export REF OBJ CFLAGS INC
echo "$SRCS" | tr ' ' '\n' | xargs -P "$JOBS" -I{} bash -c 'compile_one {}'

BLASDIR=$(dirname "$BLASLIB")
gcc -shared -fopenmp -o "$OUT/libcholmod_ref.so" "$OBJ"/*.o \
    "$BLASLIB" -Wl,-rpath,"$BLASDIR" -lm -ldl -lpthread

# flat-array helper (our code, compiled against the reference headers in place)
gcc -O2 -fPIC -fopenmp -std=c11 -Wall $INC -DBLAS32 -shared \
    -o "$OUT/libref_helper.so" "$HERE/ref_helper.c" \
    -L"$OUT" -lcholmod_ref -Wl,-rpath,'$ORIGIN' -Wl,-rpath,"$BLASDIR" -lm
echo "build_ref: OK -> $OUT/libcholmod_ref.so $OUT/libref_helper.so"
