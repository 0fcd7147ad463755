"""zerovox.cpp_amd — MI355X-native ZeroVox TTS hot path.

The directory name contains a dot (it mirrors the reference's repo name), so it is loaded through
`__graft_entry__.load_package()` under the module name `zerovox_cpp_amd`.
"""
from . import convert, gguf, sharding, synth  # noqa: F401
