# Generates oracle/_ref/include/ftk/config.hh from the reference's own template
# (/root/reference/include/ftk/config.hh.in, consumed by /root/reference/CMakeLists.txt:355-359)
# with every optional dependency OFF -- the reference's default configuration.
# Run as: cmake -DREF=/root/reference -DOUT=<dir> -P gen_config.cmake
# -DFTK_HAVE_HIP=1 (oracle/Makefile, shim target): the same generation from the PATCHED template of patches/ftk-xl-hip.patch, with the
# one option that patch adds switched on.
# This is the ONLY use of cmake: the reference's build system (its CMakeLists.txt) is never run.
file(READ "${REF}/version.txt" FTK_VERSION)
string(STRIP "${FTK_VERSION}" FTK_VERSION)
set(FTK_FP_PRECISION 32768)     # CMakeLists.txt:355 default
set(FTK_CP_MAX_NUM_VARS 3)      # CMakeLists.txt:356 default
configure_file("${REF}/include/ftk/config.hh.in" "${OUT}/include/ftk/config.hh")
