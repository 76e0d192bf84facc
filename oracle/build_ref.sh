#!/usr/bin/env bash
# Builds the UNMODIFIED reference (pcdslab/PHYNGSC) from the sources where they lie under
# /root/reference into oracle/_ref/phyNGSC.  Test infrastructure only: the binary is used to
# generate golden vectors (tests/golden/make_golden.py) and, optionally, as the "reference"
# CPU baseline.  Nothing from /root/reference is copied into this repository; object files and
# the binary go to oracle/_ref/ (git-ignored, but shipped to the GPU box by gpurun).
#
# Toolchain: g++ + the MPICH 3.3.2 that ships in this image under /opt/conda (mpi.h, libmpi.so.12).
# The mpicxx wrapper's default compiler is absent, hence -cxx=g++; the wrapper's link line picks
# conda's old libstdc++, so the final link is done with plain g++ against libmpi.so.12.
set -euo pipefail
REF=${NGSC_REFERENCE_DIR:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
MPICXX=${MPICXX:-/opt/conda/bin/mpicxx}
MPILIB=${MPILIB:-/opt/conda/lib/libmpi.so.12}
if [ ! -d "$REF" ]; then echo "build_ref: $REF not present (GPU box?) - keeping prebuilt $OUT" >&2; exit 0; fi
if [ ! -x "$MPICXX" ] || [ ! -e "$MPILIB" ]; then echo "build_ref: MPICH not found - reference unbuildable here" >&2; exit 0; fi
mkdir -p "$OUT"
for f in phyNGSC bit_stream huffman tasks; do
  "$MPICXX" -cxx=g++ -O3 -m64 -w -fopenmp -std=c++11 -I"$REF" -c "$REF/$f.cpp" -o "$OUT/$f.o"
done
g++ -O3 -m64 -fopenmp -o "$OUT/phyNGSC" "$OUT"/phyNGSC.o "$OUT"/bit_stream.o "$OUT"/huffman.o "$OUT"/tasks.o \
    -static-libstdc++ -static-libgcc "$MPILIB" -Wl,-rpath,"$(dirname "$MPILIB")"
rm -f "$OUT"/*.o
echo "build_ref: built $OUT/phyNGSC"
