# TMACConfig.cmake — find_package(TMAC) for the MI355X library, exporting what the reference's package exports
# (cmake/TMACConfig.cmake.in:65-74 of microsoft/T-MAC) so that a build which consumes that package — the llama.cpp
# fork with -DGGML_TMAC=ON — switches by pointing CMAKE_PREFIX_PATH / TMAC_DIR at this directory:
#
#   t_mac_no_tvm          INTERFACE target: include/ (t-mac/tmac_gemm_wrapper.h, t-mac/kernels.h, tmac_hip.h),
#                         links libtmac_hip.so, defines TMAC_KCFG_FILE when a kcfg.ini is known
#   TMAC_KERNELS_SOURCE   the file the consumer adds to its own sources (the reference: generated kernels.cc).  Here
#                         the kernels live in the shared library, so this is a translation unit with no code.
#   TMAC_INCLUDE_DIRS, TMAC_LIB_DIR
#
# Optional input: -DTMAC_KCFG=<path to the kcfg.ini the model was converted with> (else $TMAC_KCFG_FILE at run time).
get_filename_component(_tmac_root "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
set(TMAC_INCLUDE_DIRS "${_tmac_root}/include")
set(TMAC_LIB_DIR "${_tmac_root}/tmac_amd/lib")

find_library(tmac_hip_LIBRARY tmac_hip HINTS "${TMAC_LIB_DIR}" NO_DEFAULT_PATH)
if(NOT tmac_hip_LIBRARY)
  set(TMAC_FOUND FALSE)
  set(TMAC_NOT_FOUND_MESSAGE "libtmac_hip.so not found in ${TMAC_LIB_DIR}: run `make -C ${_tmac_root}/tmac_amd/csrc` (hipcc, gfx950)")
  return()
endif()

set(TMAC_COMPILE_DEFS "")
if(TMAC_KCFG)
  get_filename_component(_tmac_kcfg "${TMAC_KCFG}" ABSOLUTE)
  set(TMAC_COMPILE_DEFS "TMAC_KCFG_FILE=${_tmac_kcfg}")
endif()

if(NOT TARGET t_mac_no_tvm)
  add_library(t_mac_no_tvm INTERFACE IMPORTED)
  set_target_properties(t_mac_no_tvm PROPERTIES
    INTERFACE_INCLUDE_DIRECTORIES "${TMAC_INCLUDE_DIRS}"
    INTERFACE_LINK_LIBRARIES "${tmac_hip_LIBRARY}"
    INTERFACE_COMPILE_DEFINITIONS "${TMAC_COMPILE_DEFS}"
    INTERFACE_COMPILE_FEATURES cxx_std_17)
endif()
set(TMAC_KERNELS_SOURCE "${CMAKE_CURRENT_LIST_DIR}/tmac_kernels_shim.cc")
set(TMAC_FOUND TRUE)
