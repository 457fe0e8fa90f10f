/*
 * volxel_brick.h -- C ABI of the native brick-layout producer (part of libvolxel_hip.so).
 *
 * Stands in for the reference's Rust/wasm `dicom_preprocessor` crate at its wasm-bindgen
 * boundary.  Rust is not available in the build image, so the
 * producer is C++ (volxel_amd/csrc/brick_builder.cpp); the entry points mirror the
 * exported functions and BrickGrid getters one to one:
 *
 *   read_dicoms_to_grid(Vec<Uint8Array>) -> BrickGrid     dicom_preprocessor/src/lib.rs:193-202
 *   BrickGrid getters                                     dicom_preprocessor/src/brick.rs:273-363
 *   grid.free()                                           volxel-3d-viewer/src/worker.ts:54
 *
 * This part runs on host cores (like the reference's: a Web Worker) and needs no GPU.
 */
#ifndef VOLXEL_BRICK_H
#define VOLXEL_BRICK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct VxBrickGrid VxBrickGrid;

#define VXB_OK 0
#define VXB_ERR_INVALID 1
#define VXB_ERR_TOO_MANY_BRICKS 2 /* the "Exceeded max brick count" panic, brick.rs:79-81 */

/*
 * BrickGrid::construct (brick.rs:76-204) over a stacked u16 volume, i.e. what
 * read_dicoms_internal (lib.rs:142-191) hands to it: voxels x-fastest
 * (buf3d.rs:26-28), density = raw / max_value (dicom.rs:7-17), spacing = PixelSpacing x,y
 * and SliceThickness (lib.rs:105-124,138).  max_value == 0 means "maximum over the data"
 * (lib.rs:92-101,167-169).  n_threads <= 0: all host cores.
 */
int vxb_build_from_u16(const uint16_t* voxels, const uint32_t dims[3], const float spacing[3],
                       uint16_t max_value, int n_threads, VxBrickGrid** out);
/*
 * read_dicoms_to_grid (lib.rs:193-202): n_files Part-10 DICOM buffers (uncompressed, 16-bit unsigned
 * monochrome, implicit or explicit VR little endian), stacked in the order given
 * (lib.rs:142-191), then BrickGrid::construct.  Errors carry the reference's panic messages
 * ("Currently only 16bit samples are supported", ...).
 */
int vxb_read_dicoms_to_grid(const uint8_t* const* files, const uint64_t* sizes, uint32_t n_files,
                            int n_threads, VxBrickGrid** out);
void vxb_free(VxBrickGrid* g);
const char* vxb_last_error(void);

/* brick.rs:275-303 ind_* / range_* / atlas_*  */
void vxb_indirection_size(const VxBrickGrid* g, uint32_t out[3]);
void vxb_range_size(const VxBrickGrid* g, uint32_t out[3]);
void vxb_atlas_size(const VxBrickGrid* g, uint32_t out[3]);
/* brick.rs:354-362: views stay valid until vxb_free */
const uint32_t* vxb_indirection_data(const VxBrickGrid* g);
const uint16_t* vxb_range_data(const VxBrickGrid* g); /* LE u16 pairs [max,min]           */
const uint8_t* vxb_atlas_data(const VxBrickGrid* g);
/* brick.rs:338-352 */
uint32_t vxb_range_mipmaps(const VxBrickGrid* g);
const uint16_t* vxb_range_mipmap(const VxBrickGrid* g, uint32_t index);
void vxb_range_mipmap_stride(const VxBrickGrid* g, uint32_t index, uint32_t out[3]);
/* brick.rs:305-307,322-336 */
void vxb_transform(const VxBrickGrid* g, float out16[16]);
float vxb_minorant(const VxBrickGrid* g);
float vxb_majorant(const VxBrickGrid* g);
void vxb_index_extent(const VxBrickGrid* g, uint32_t out[3]);
/* brick.rs:309-320 (histogram of raw values, lib.rs:87-102; gradient dicom.rs:39-66) */
uint32_t vxb_histogram_len(const VxBrickGrid* g);
const uint32_t* vxb_histogram(const VxBrickGrid* g);
const int32_t* vxb_histogram_gradient(const VxBrickGrid* g);
uint32_t vxb_histogram_gradient_min(const VxBrickGrid* g);
uint32_t vxb_histogram_gradient_max(const VxBrickGrid* g);
/* number of non-constant bricks allocated in the atlas (brick.rs:127-128) */
uint32_t vxb_brick_counter(const VxBrickGrid* g);
/* Grid::lookup on the built grid (brick.rs:208-230) */
float vxb_lookup(const VxBrickGrid* g, uint32_t x, uint32_t y, uint32_t z);

#ifdef __cplusplus
}
#endif
#endif /* VOLXEL_BRICK_H */
