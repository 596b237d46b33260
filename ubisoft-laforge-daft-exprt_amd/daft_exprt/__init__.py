"""MI355X-native Daft-Exprt: the reference's Python surface (`DaftExprt`, `DaftExprtLoss`,
`HyperParams`, trainer / generator entry points) on top of hand-written gfx950 HIP kernels
(`csrc/` -> `libdaftexprt_hip.so`, C ABI in `include/daft_exprt_hip.h`)."""
