#!/bin/bash
# AddressSanitizer + UBSan over the device code, without a device (host code only: no GPU sanitizer is involved).
# The CPU build of the engine (engine.hip's host code + every kernel, lane by lane on fibers) and the schedule emulator
# are rebuilt with -fsanitize=address,undefined and the CPU tests that drive them run under the sanitizer runtime:
# "device" memory is malloc'ed host memory there, so a kernel or a schedule entry that reaches outside Lx, Dinv, the
# relative maps, X or an LDS array is reported with the kernel's source line -- the same access would be a GPU memory
# fault on the box.  The regular libraries are restored afterwards.      usage: bash tools/asan_cpuemu.sh [pytest args]
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
H=$R/tests/helpers
CXX=/opt/rocm/lib/llvm/bin/clang++
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
# (-fno-sanitize=function: the engine calls librccl through pointers it declares itself -- ints for the enums, its own 128-byte struct for the
# id -- and the stand-in of tests/helpers/hipcpu declares the same entry points with ITS types: the same C ABI, different C++ types)
SAN="-fsanitize=address,undefined -fno-sanitize=function -fno-sanitize-recover=undefined -shared-libasan -g -fno-omit-frame-pointer"
mkdir -p /tmp/asan_keep && cp $H/libmi355_cpuemu.so $H/libsched_emu.so /tmp/asan_keep/
restore() { cp /tmp/asan_keep/libmi355_cpuemu.so /tmp/asan_keep/libsched_emu.so $H/; }
trap restore EXIT
C=$R/suitesparse_amd/csrc
$CXX -x c++ -std=c++17 -O1 $SAN -w -pthread -fPIC -shared -DHIPCPU_FAKE_RCCL -I $H/hipcpu $C/engine.hip $C/schedule.cpp $C/symbolic.cpp -o $H/libmi355_cpuemu.so || exit 1
$CXX -std=c++17 -O1 $SAN -w -pthread -fPIC -shared -I $H/hipcpu $H/sched_emu.cpp $C/schedule.cpp -o $H/libsched_emu.so || exit 1
cd $R
# (tests/test_cholmod_dropin.py is left out: it loads a library with RTLD_DEEPBIND, which the sanitizer runtime refuses)
LD_PRELOAD="$RT${LD_PRELOAD:+:$LD_PRELOAD}" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:halt_on_error=1 \
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
python -m pytest ${ASAN_TESTS:-tests/test_abi_on_cpu.py tests/test_multi_abi.py tests/test_device_source_on_cpu.py tests/test_multigpu.py \
    tests/test_schedule.py tests/test_fuzz_schedule.py tests/test_panel_groups.py} -q -m "not gpu" -p no:cacheprovider "$@"
