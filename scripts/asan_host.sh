# CPU-side sanitizer run (GPU ASan is not available on this pool): builds the library's host code with
# -fsanitize=address,undefined and drives the GGUF loader / WAV writer with truncated and corrupted files.
set -e
cd "$(dirname "$0")/.."
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
make -s -C zerovox.cpp_amd/csrc -j8 OUT=../../variants/libzv_asan.so BIN= OBJDIR=../../variants/_asan \
     EXTRA="-fsanitize=address,undefined -fno-omit-frame-pointer -g -Wno-option-ignored" ../../variants/libzv_asan.so
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -fsanitize=address,undefined -shared-libasan tests/native/host_fuzz.cpp -o variants/host_fuzz \
     -Lvariants -l:libzv_asan.so -Wl,-rpath,$PWD/variants -Wl,-rpath,$(dirname $RT)
T=$(mktemp -d)
python - "$T" <<'PY'
import sys
sys.path.insert(0, '.')
from __graft_entry__ import load_package
load_package()
from zerovox_cpp_amd import synth
synth.write_checkpoint(sys.argv[1] + "/tiny.gguf", synth.TINY, 5)
PY
ASAN_OPTIONS=detect_leaks=1 timeout 300 variants/host_fuzz "$T/tiny.gguf" "$T"
rm -rf "$T"
