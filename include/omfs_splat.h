/*
 * omfs_splat.h -- C ABI of libomfs_splat.so: the MI355X (gfx950) engine behind the
 * reference's process boundary.
 *
 * What this replaces.  The reference has no FFI: `02_Visual_Engine/train_ghost.py:227-271`
 * and `02_Visual_Engine/render_surgery.py:289-315` spawn `gaussian_avatars_repo/train.py`
 * and `.../render.py` (un-vendored, `.gitignore:27`), whose per-iteration work is the
 * FLAME-mesh-bound Gaussian splat.  Each entry point below is one stage of that absent
 * engine (SURVEY.md §8a row a-13, §8b "inner contract"); the Python engine
 * (`omfs_4d_video_gen_amd/engine/`) binds them with ctypes and is what `train.py` /
 * `render.py` in that directory run.  `omfs_simpleflame_*` replace the tensor math of
 * `02_Visual_Engine/flame_fitter.py:154-197` (forward) and `:377-413` (fit loop).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller unless the name ends in _host;
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing syncs;
 *  - no hidden allocation, no global state; scratch comes in through the structs;
 *  - return 0 on success, a negative OMFS_ERR_* otherwise, text via omfs_last_error()
 *    (thread-local);
 *  - fp32 throughout; Gaussian parameters are a planar SoA `params[59][n_pad]`
 *    (plane order: OMFS_P_* below), n_pad a multiple of 256.
 */
#ifndef OMFS_SPLAT_H
#define OMFS_SPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMFS_ABI_VERSION 8
#define OMFS_TILE 16
#define OMFS_SEG 128     /* list entries per backward segment                                          */
#define OMFS_NPLANES 59
#define OMFS_P_XYZ 0      /* 3 planes: local position in the parent triangle frame          */
#define OMFS_P_SCALE 3    /* 3 planes: log scale (triangle-relative)                         */
#define OMFS_P_ROT 6      /* 4 planes: quaternion w,x,y,z (un-normalised)                    */
#define OMFS_P_OPACITY 10 /* 1 plane : opacity logit                                         */
#define OMFS_P_SH 11      /* 48 planes: SH coefficient k, channel c at 11 + 3*k + c          */
#define OMFS_FLAME_JOINTS 5
#define OMFS_FLAME_POSEDIRS 36

#define OMFS_OK 0
#define OMFS_ERR_ARG (-1)
#define OMFS_ERR_HIP (-2)
#define OMFS_ERR_CAPACITY (-3)

/* status word bits written by the device (omfs_raster_buffers.status) */
#define OMFS_STATUS_DUP_OVERFLOW 1u  /* sum of tiles touched exceeded dup_capacity: frame left empty */

int omfs_abi_version(void);
const char* omfs_last_error(void);

/* ------------------------------------------------------------------ FLAME rig -> vertices
 * Replaces (and completes: SimpleFLAME has no LBS) flame_fitter.py:154-197 for the engine. */
typedef struct omfs_flame_rig {
  int n_verts;              /* V                                                               */
  int v_pad;                /* V rounded up to 16                                              */
  int n_expr;               /* expression coefficients used per frame (<=100)                  */
  int k_pad;                /* n_expr + 36 rounded up to 16                                    */
  const float* basis_tiled; /* [3][v_pad/16][k_pad/16][64][4]  MFMA-A tiles, see DESIGN.md     */
  const float* v_static;    /* [3][v_pad]   template + shape blendshapes + static_offset       */
  const float* lbs_weights; /* [v_pad][8]   5 skinning weights + 3 zero pads                   */
  const float* j_static;    /* [5][3]       J_regressor . v_static                             */
  const float* j_expr;      /* [15][n_expr] J_regressor . exprdirs                             */
} omfs_flame_rig;

/* Joint transforms + blendshape coefficients for `n_frames` frames.
 * expr [n_frames][n_expr], rotmats [n_frames][5][9] row-major, translation unused here.
 * Writes joint_xf [n_frames][5][12] (R row-major 9, t 3) and coef [k_pad][b_pad]
 * (b_pad = n_frames rounded up to 16; rows: expr, then 36 pose features, then zeros).
 * frame_index (device, may be NULL): batch column b shows row frame_index[b] of expr / rotmats (and of translation /
 * dynamic_offset in omfs_flame_lbs) instead of row b -- an arbitrary set of timesteps in one launch. */
int omfs_flame_joints(const omfs_flame_rig* rig, const float* expr, const float* rotmats,
                      int n_frames, float* joint_xf, float* coef, const int32_t* frame_index, void* stream);
/* The same from axis-angle poses: pose [*][15] (global, neck, jaw, eye-L, eye-R; rows as expr).  The rotation matrices of
 * the posed rows are computed in the same launch (omfs_flame_rodrigues's formula) and stored to rotmats [*][45]:
 * FLAME fine-tuning poses from the current parameters without a separate Rodrigues launch. */
int omfs_flame_joints_pose(const omfs_flame_rig* rig, const float* expr, const float* pose, float* rotmats,
                           int n_frames, float* joint_xf, float* coef, const int32_t* frame_index, void* stream);

/* verts [n_frames][v_pad][4] = LBS(v_static + basis . coef) (+ dynamic_offset) + translation.
 * dynamic_offset may be NULL; layout [n_frames][V][3]. translation [n_frames][3].
 * v_shaped_out (may be NULL): [n_frames][v_pad][4] the blend-shaped vertices in front of the skinning (kept for
 * omfs_flame_skin_bwd). */
int omfs_flame_lbs(const omfs_flame_rig* rig, const float* coef, const float* joint_xf,
                   const float* translation, const float* dynamic_offset, int n_frames,
                   float* verts, float* v_shaped_out, const int32_t* frame_index, void* stream);

/* ONE frame, omfs_flame_joints(_pose) + omfs_flame_lbs in a single launch (the training step poses one view per iteration):
 * every 16-vertex wave evaluates the frame's joints itself while its first basis tiles are in flight -- the same code, the
 * same bits.  expr / rotmats / pose / translation / dynamic_offset point at the FRAME's row, or, with frame_index (device,
 * one entry), at row 0 of the sequence arrays.  pose == NULL: the joints come from rotmats [45]; otherwise rotmats receives
 * the frame's five matrices.  joint_xf [60] and coef [k_pad][16] (column 0) are written for the backward pass / inspection. */
int omfs_flame_pose_lbs(const omfs_flame_rig* rig, const float* expr, float* rotmats, const float* pose,
                        const float* translation, const float* dynamic_offset, float* joint_xf, float* coef,
                        float* verts, float* v_shaped_out, const int32_t* frame_index, void* stream);

/* face_xf [n_frames][n_faces][16]: R row-major (columns a0,n,a2) 9, centre 3, scale 1, pad 3 */
int omfs_face_frames(const float* verts, int v_pad, const int32_t* faces, int n_faces, int n_frames,
                     float* face_xf, void* stream);

/* ------------------------------------------------------------------ SimpleFLAME landmark model
 * Replaces the tensor math of flame_fitter.py:154-197 (SimpleFLAME.forward) and its autograd pass in
 * the fit loop :377-413.  The barycentric landmark mix (:192-195) is folded into the bases by the host:
 * lmk_template [L][3], lmk_basis [L][3][n_shape+n_expr], lmk_lower [L] (mixed lower-face mask, :178-182). */
typedef struct omfs_simpleflame {
  int n_landmarks, n_shape, n_expr;
  const float* lmk_template;
  const float* lmk_basis;
  const float* lmk_lower;
} omfs_simpleflame;

/* shape [B][n_shape], expr [B][n_expr], rotation/jaw/translation [B][3] -> landmarks [B][L][3];
 * p_scratch [B][L][3] keeps the pre-rotation points for the backward pass */
int omfs_simpleflame_fwd(const omfs_simpleflame* m, const float* shape, const float* expr, const float* rotation,
                         const float* jaw, const float* translation, int n_frames, float* landmarks,
                         float* p_scratch, void* stream);
/* dlandmarks [B][L][3] -> per-frame gradients of every input (dshape [B][n_shape]: the caller sums over
 * frames when the shape is shared); g_scratch [B][L][3] */
int omfs_simpleflame_bwd(const omfs_simpleflame* m, const float* rotation, const float* p_scratch,
                         const float* dlandmarks, int n_frames, float* g_scratch, float* dshape, float* dexpr,
                         float* drotation, float* djaw, float* dtranslation, void* stream);

/* One iteration of fit_flame_to_landmarks (flame_fitter.py:377-413) entirely on the device, no autograd: landmarks of all
 * frames (shared shape), pseudo-perspective x/(-z+1e-8) against the 2-D targets with the masked MSE of :390-392, the L2
 * regularisers (:395-397) and the temporal smoothness terms (:400-404) in closed form, then torch.optim.Adam semantics on the
 * five tensors (order everywhere: shape, expr, rotation, jaw, translation).  loss_out receives the loss BEFORE the update.
 * scratch: omfs_flame_fit_scratch_floats(m, n_frames) floats.                                                        */
typedef struct omfs_flame_fit {
  int n_frames, n_use;       /* frames; landmarks that enter the loss (min(68, n_landmarks))                          */
  const float* target;       /* [n_frames][n_use][2] in [-1,1]                                                        */
  const float* valid;        /* [n_frames] 1 = frame has landmarks                                                    */
  float inv_denom;           /* 1 / (valid frames * n_use)                                                            */
  float lr[5], beta1, beta2, eps;
  int step;                  /* 1-based iteration (bias correction)                                                   */
  float *shape, *expr, *rotation, *jaw, *translation;   /* [n_shape], [T][n_expr], [T][3] x3: updated in place        */
  float *m[5], *v[5];        /* Adam moments, same shapes                                                             */
  float* scratch;
  float* loss_out;           /* [1]                                                                                   */
} omfs_flame_fit;
size_t omfs_flame_fit_scratch_floats(const omfs_simpleflame* m, int n_frames);
int omfs_flame_fit_step(const omfs_simpleflame* m, const omfs_flame_fit* f, void* stream);

/* ------------------------------------------------------------------ rasteriser */
typedef struct omfs_camera {
  float view[12];   /* world->view, rows of [R|t] (3x4 row-major)                               */
  float cam_pos[3];
  float fx, fy;
  float cx, cy;     /* (width-1)/2, (height-1)/2                                                */
  float limx, limy; /* 1.3*tan(fov/2)                                                           */
  int width, height;
  int sh_degree;
  float bg[3];
} omfs_camera;

typedef struct omfs_gaussians {
  int n;                  /* Gaussians                                                          */
  int n_pad;              /* plane stride (multiple of 256)                                     */
  const float* params;    /* [59][n_pad]                                                        */
  const int32_t* binding; /* [n] parent face                                                    */
} omfs_gaussians;

typedef struct omfs_raster_buffers {
  /* per Gaussian (written by project_fwd): the projected-splat record, three float4.  omfs_record_stride() == 1 (the shipped
   * build): three planar [n][4] arrays.  A library built with -DOMFS_REC_STRIDE=4 reads ONE 64-byte record per Gaussian instead
   * (rec [n][16]: g1 = g0 + 4, g2 = g0 + 8 floats; measured 0.3 % slower per iteration: csrc/common.hpp) and returns 4;
   * omfs_project_fwd refuses pointers that do not match the layout it was built for. */
  float* g0;              /* [n][4] mean2d.x, mean2d.y, conic.a, conic.b                         */
  float* g1;              /* [n][4] conic.c, opacity, r, g                                       */
  float* g2;              /* [n][4] b, depth, bits(radius | clamp<<28), bits(rect x0|y0<<8|x1<<16|y1<<24) */
  /* binning */
  uint32_t* tile_count;   /* [n_tiles]                                                           */
  uint32_t* tile_start;   /* [n_tiles+1] exclusive scan; [n_tiles] = D                           */
  uint32_t* tile_cursor;  /* [n_tiles] scratch                                                   */
  uint32_t* tile_order;   /* [n_tiles] tiles by descending count bucket                          */
  uint32_t* keys;         /* [dup_capacity][2] (depth bits, gaussian id), tile-segmented         */
  uint32_t* keys_tmp;     /* [dup_capacity][2] scratch for tiles longer than the LDS capacity    */
  uint32_t* sorted_ids;   /* [dup_capacity] per-tile front-to-back Gaussian ids                  */
  uint32_t dup_capacity;
  uint32_t sort_lds_pairs; /* longest tile list whose bucket-ordered copy is kept in LDS (0 = default 7936 pairs = 78 KB,
                              two workgroups per CU; 8 B of LDS each); longer lists keep it in keys_tmp        */
  uint32_t* status;       /* [2] word 0: OMFS_STATUS_* bits, OR-ed by kernels (caller zeroes); word 1: stamp of the
                             (Gaussians, camera) whose tile-test ballots keys_tmp holds (library-owned, zero initially) */
  /* forward checkpoints for the depth-parallel backward pass: per pixel (T, C.rgb) on entering list segment k
   * (k >= 1) of tile t, stored at slot tile_start[t]/OMFS_SEG + t + k; seg_capacity >= n_tiles + dup_capacity/OMFS_SEG */
  float* seg_ckpt;        /* [seg_capacity][256][4]                                               */
  uint32_t* order_seg0;   /* [n_tiles+1] segments owned by the tiles before each launch-order position     */
  uint32_t seg_capacity;
  /* per pixel */
  float* image;           /* [3][height][width]                                                  */
  float* final_T;         /* [height][width]                                                     */
  uint32_t* n_contrib;    /* [height][width]                                                     */
  uint32_t flags;         /* OMFS_RB_FORWARD_ONLY: no backward pass will follow (render_surgery): the forward skips
                             the segment checkpoints (only the hand-over slots between its two kernels are written) */
  uint32_t* n_visible;    /* optional [1]: number of Gaussians with radius > 0 in this view; omfs_project_fwd clears it,
                             omfs_bin_count accumulates it (what omfs_count_visible computes, without its two
                             dispatches); may be NULL                                                              */
  uint32_t* quad_depth;   /* optional [n_tiles][4] (ABI 7): deepest last contributor of every (tile, 8x8 quadrant), written by
                             omfs_composite_fwd in training mode and read by omfs_composite_bwd (NULL: the library keeps the
                             table inside `keys`).  A caller that hands in a buffer of its own PER VIEW and leaves it alone
                             between two visits of the view also gives the forward a HINT: the few quadrants that went deep
                             last time are walked at raised wave priority (speed only: results do not depend on the content,
                             which may be anything -- zero it once).                                                    */
} omfs_raster_buffers;
#define OMFS_RB_FORWARD_ONLY 1u
#define OMFS_RB_NO_DEPTH_HINT 2u   /* omfs_composite_fwd ignores the priority hint of a caller-owned quad_depth table (A/B measurements) */

int omfs_record_stride(void);   /* float4 units between two Gaussians' records in g0 / g1 / g2: 1 (planar, the shipped build) or 4 (one 64-byte record) */
/* deform + project + colour for one view -> g0,g1,g2. face_xf [n_faces][16]. */
int omfs_project_fwd(const omfs_gaussians* g, const float* face_xf, const omfs_camera* cam,
                     const omfs_raster_buffers* rb, void* stream);
/* the four binning steps, separately launchable (omfs_bin_sort = all four in order):
 *  count   : tile_count[t] = #Gaussians whose 3-sigma rectangle contains tile t AND that can reach
 *            alpha >= 1/255 at one of its pixel centres (frozen test, DESIGN.md "Binning")
 *  scan    : tile_count -> tile_start (exclusive scan), tile_order, zeroed tile_cursor; tile_count is consumed (left
 *            zeroed for the next view: it must be zero before the first omfs_bin_count)
 *  scatter : (depth bits, id) pairs into their tile's segment of keys; REPLAYS the tile-test outcomes that the
 *            omfs_bin_count of the same view recorded in keys_tmp (one 64-bit ballot per walk step, followed by every
 *            workgroup's list of non-empty (tile, count) pairs, from which the slot ranges are taken) when status[1]
 *            still carries that call's stamp (same g->n, g->params and camera, no omfs_tile_sort and no omfs_project_fwd
 *            in between: a projection writes new records and clears the stamp); in any other call order the test is
 *            re-evaluated and the pairs are counted again (slower, same result)
 *  sort    : per-tile sort by (depth bits, id) -> sorted_ids (keys_tmp is scratch again from here on)       */
int omfs_bin_count(const omfs_gaussians* g, const omfs_camera* cam, const omfs_raster_buffers* rb, void* stream);
int omfs_bin_scan(const omfs_camera* cam, const omfs_raster_buffers* rb, void* stream);
int omfs_bin_scatter(const omfs_gaussians* g, const omfs_camera* cam, const omfs_raster_buffers* rb, void* stream);
int omfs_tile_sort(const omfs_camera* cam, const omfs_raster_buffers* rb, void* stream);
int omfs_bin_sort(const omfs_gaussians* g, const omfs_camera* cam, const omfs_raster_buffers* rb, void* stream);
/* front-to-back alpha composite -> image, final_T, n_contrib */
int omfs_composite_fwd(const omfs_camera* cam, const omfs_raster_buffers* rb, void* stream);
/* image [3][H][W] fp32 -> rgb8 [H][W][3], clamp to [0,1], round(x*255) */
int omfs_image_to_rgb8(const float* image, int width, int height, uint8_t* rgb8, void* stream);
/* [H][W][3] uint8 -> [3][H][W] fp32 (value / 255): training targets stored as 8-bit RGB in HBM */
int omfs_rgb8_to_image(const uint8_t* rgb8, int width, int height, float* image, void* stream);
/* the same conversion as PNG scanlines: rows [height][1 + 3*width], byte 0 of every row is the filter type 0 */
int omfs_image_to_png_rows(const float* image, int width, int height, uint8_t* rows, void* stream);

/* ------------------------------------------------------------------ backward + optimiser */
typedef struct omfs_grad_buffers {
  float* dsplat;          /* [n][16] per-Gaussian record atomically accumulated by omfs_composite_bwd: the moments
                             S_x, S_y, S_xx, S_xy, S_yy of dL/dG*G over the pixels (dx = mean - pixel; omfs_project_bwd
                             turns them into d mean2d / d conic), dopacity, drgb; zero before the first use -- omfs_project_bwd
                             clears every record it consumes, so a composite_bwd / project_bwd pair leaves it zeroed */
  float* grads;           /* [59][n_pad] parameter gradients (overwritten)                         */
  const float* dimage;    /* [3][H][W] dL/dimage                                                   */
  float* densify_stats;   /* optional [2][n_pad]: += |d mean2d| in NDC-scaled units (x W/2, y H/2) and += 1
                             for every Gaussian visible in this view (adaptive density control); may be NULL */
  float* dface;           /* optional [n][16]: each Gaussian's contribution to dL/d(frame record of its parent triangle)
                             -- R row-major (9), centre (3), scale (1) -- overwritten, no atomics (FLAME fine-tuning:
                             omfs_face_frames_bwd sums them per triangle); may be NULL                            */
  float* drgb_out;        /* optional [3][n_pad]: when set, omfs_project_bwd writes dL/d(colour) of this view (zero for
                             clamped channels and invisible Gaussians) here and leaves the 45 SH planes of degree >= 1
                             (planes 14..58) untouched: data-parallel ranks exchange these 3 planes instead of 45 and
                             rebuild the summed gradient with omfs_sh_rest_grads; may be NULL                     */
  float* dir_out;         /* optional [3][n_pad] (ABI 7; only with drgb_out): the unit view direction camera -> Gaussian that the
                             colour was evaluated with (0 for invisible Gaussians).  With drgb_out it is everything the gradient
                             of the 45 SH planes of degree >= 1 is made of, Y_k(dir) * drgb[c]: omfs_adam_step_sh_rest forms
                             it where it is consumed, so those planes are never written nor read; may be NULL          */
  long long* dsplat_fx;   /* optional [n][16] (ABI 8), zero before the first use: DETERMINISTIC accumulation.  omfs_composite_bwd then
                             adds every contribution as a 64-bit fixed-point integer (columns 0..4 scaled by 2^38, 5..8 by 2^46;
                             integer addition is associative, so the sums do not depend on the order the waves arrive in),
                             converts the totals into `dsplat` and leaves this buffer zero again.  Two runs of the same training
                             are then bit-identical (with omfs_face_frames_bwd_fx and the split FLAME backward, which has no
                             float atomics); results differ from the float-atomic path by the quantisation (<= 2^-39 resp.
                             2^-47 per contribution) and saturate beyond +-2^24 resp. +-2^16.  A few us slower.  May be NULL. */
  uint32_t n_records;     /* with dsplat_fx: the number of Gaussians the records were projected for (rows of dsplat / dsplat_fx
                             the conversion walks); ignored otherwise                                                      */
} omfs_grad_buffers;

/* Must follow omfs_composite_fwd of the SAME lists run in training mode (flags without OMFS_RB_FORWARD_ONLY), with `keys`
 * untouched in between: besides the checkpoints the forward leaves the backward's work tables there (ABI 7: segment -> (tile,
 * k) and, unless the caller passes rb->quad_depth, the quadrant depths), which the next frame's binning overwrites.        */
int omfs_composite_bwd(const omfs_camera* cam, const omfs_raster_buffers* rb, const omfs_grad_buffers* gb, void* stream);

/* FLAME fine-tuning (upstream GaussianAvatars optimises the per-timestep FLAME parameters together with the
 * Gaussians; SURVEY.md section 8b names flame_lbs_bwd in the inner contract).  One frame at a time:
 *   omfs_flame_rodrigues : axis-angle [n][3] -> rotation matrices [n][9] (the formula of flame_fitter.py:133-152)
 *   omfs_face_frames_bwd : per-Gaussian records dface [n][16] (from omfs_project_bwd), summed per triangle through the
 *                          CSR (face_start [F+1], face_gauss [n]: Gaussians sorted by parent triangle) -> dverts
 *                          [v_pad][4] += (atomics; caller zeroes)
 *   omfs_flame_skin_bwd  : dverts (CONSUMED: every row read is left zeroed, so the next frame's omfs_face_frames_bwd
 *                          needs no clearing pass) -> dv_shaped [V][3] (gradient of the blend-shaped vertices, overwritten) and
 *                          sums [omfs_flame_skin_rows(rig)][64]: one row of partial sums per wave,
 *                          { d joint_xf [5][12], d translation [3], pad } (overwritten, no atomics); v_shaped
 *                          [v_pad][4] is the optional second output of omfs_flame_lbs, joint_xf [60] that of
 *                          omfs_flame_joints for the frame
 *   omfs_flame_param_bwd : dv_shaped, sums (the rows above) -> d expr [n_expr], d pose [5][3] (axis-angle: global,
 *                          neck, jaw, eyes), d translation [3]; dcoef [n_coef + 1] is scratch whose LAST word is a
 *                          block ticket that must be zero before the first call (the call leaves it zero): the basis^T
 *                          product and the serial front run in ONE launch, the block that finishes last does the front.
 *   omfs_flame_skin_param_bwd : omfs_flame_skin_bwd + omfs_flame_param_bwd in ONE launch (ABI 6; what the trainer calls): a
 *                          workgroup owns 32 vertices, forms their dv_shaped in LDS, multiplies it into the TRANSPOSED basis
 *                          basis_t [3 V][n_coef] (row 3 v + c; contiguous per wave, no cross-lane reduction) and adds its share
 *                          of dcoef and of the 63 joint / translation sums with float atomics (into one of 16 copies of the
 *                          accumulators: adds to one cache line serialise); the workgroup that finishes last (ticket) adds
 *                          the copies up and runs the serial front.  dcoef holds omfs_flame_skin_param_scratch_floats(0)
 *                          floats, sums omfs_flame_skin_param_scratch_floats(1); both must be ZERO before the first call,
 *                          every call leaves them zero.
 *   omfs_adam_flat       : torch.optim.Adam step on a flat buffer (the FLAME parameter tensors)
 *   omfs_adam_flat_multi : the same step on up to 4 flat tensors in one launch (host arrays of n_tensors device pointers,
 *                          sizes and learning rates); the gradients are CONSUMED (zeroed once read), so dense gradient
 *                          tensors of which one row is written per step never need clearing */
int omfs_face_frames_bwd(const float* verts, int v_pad, const int32_t* faces, int n_faces, const float* dface,
                         const int32_t* face_start, const int32_t* face_gauss, float* dverts, void* stream);
/* The same with DETERMINISTIC accumulation (ABI 8): the per-corner contributions are added into dverts_fx [v_pad][4] as 64-bit
 * fixed-point integers (scale 2^40; zero before the first call, left zero), then converted into dverts (overwritten rows:
 * those of vertices that received a contribution).  See omfs_grad_buffers.dsplat_fx. */
int omfs_face_frames_bwd_fx(const float* verts, int v_pad, const int32_t* faces, int n_faces, const float* dface,
                            const int32_t* face_start, const int32_t* face_gauss, float* dverts, long long* dverts_fx, void* stream);
int omfs_flame_skin_bwd(const omfs_flame_rig* rig, const float* v_shaped, const float* joint_xf, float* dverts,
                        float* dv_shaped, float* sums, void* stream);
int omfs_flame_rodrigues(const float* axis_angle, int n, float* rotmats, void* stream);
int omfs_flame_skin_rows(const omfs_flame_rig* rig);
int omfs_flame_param_bwd(const omfs_flame_rig* rig, const float* basis_dense, int n_coef, const float* dv_shaped,
                         const float* expr, const float* pose, const float* sums, float* dcoef, float* dexpr, float* dpose,
                         float* dtrans, void* stream);
int omfs_flame_skin_param_scratch_floats(int which);
int omfs_flame_skin_param_bwd(const omfs_flame_rig* rig, const float* basis_t, int n_coef, const float* v_shaped,
                              const float* joint_xf, float* dverts, const float* expr, const float* pose, float* dcoef,
                              float* sums, float* dexpr, float* dpose, float* dtrans, void* stream);
int omfs_adam_flat(float* params, const float* grads, float* m, float* v, int n, float lr, float beta1, float beta2,
                   float eps, int step, float grad_scale, void* stream);
struct omfs_step_state;
int omfs_adam_flat_multi(int n_tensors, float* const* params_host, float* const* grads_host, float* const* m_host,
                         float* const* v_host, const int* n_host, const float* lr_host, float beta1, float beta2, float eps,
                         int step, float grad_scale, const struct omfs_step_state* state_dev, void* stream);

typedef struct omfs_reg_params {
  float lambda_xyz, thr_xyz, lambda_scale, thr_scale;
  const uint32_t* n_visible; /* device word: #visible Gaussians of this view (omfs_count_visible); the
                                regularisers are means over the visible Gaussians                   */
} omfs_reg_params;

int omfs_project_bwd(const omfs_gaussians* g, const float* face_xf, const omfs_camera* cam,
                     const omfs_raster_buffers* rb, const omfs_grad_buffers* gb,
                     const omfs_reg_params* reg, void* stream);

/* Data-parallel exchange in compact form.  The gradient of every SH coefficient is a rank-1 product per Gaussian and view:
 * d SH[k][c] = Y_k(dir) * dL/dcolour[c] (Y_0 = the constant C0).  Every rank holds the same parameters, triangle frames and
 * cameras, so it can rebuild the sum over all ranks' views from the gathered dL/dcolour alone:
 *   grads[(11 + 3 k + c) * n_pad + i] = sum_w Y_k(dir_w(i)) * drgb_all[w][c][i],   k = 0..15   (ABI 8: k = 0 included, so only
 * the 11 geometry / opacity planes travel in the all-reduce; ABI <= 7 rebuilt k = 1..15 and all-reduced 14 planes)
 * face_xf_all [n_views][n_faces][16] (the views' timesteps), cam_pos_table [*][3] indexed by views->view[w],
 * drgb_all [n_views][3][n_pad].  Identical on every rank (same order of summation). */
typedef struct omfs_view_set {
  int n_views;            /* <= 16                                                                 */
  int view[16];           /* row of cam_pos_table of each view                                      */
} omfs_view_set;

/* the same three planes omfs_project_bwd writes to drgb_out, available right after omfs_composite_bwd (so that the
 * all-gather can overlap omfs_project_bwd) */
int omfs_extract_drgb(const omfs_raster_buffers* rb, const float* dsplat, int n, int n_pad, float* drgb_out, void* stream);
int omfs_sh_rest_grads(const omfs_gaussians* g, const float* face_xf_all, int n_faces, const float* cam_pos_table,
                       const omfs_view_set* views, const float* drgb_all, int sh_degree, float* grads, void* stream);

/* The exchange itself, for hosts that are not PyTorch (SURVEY.md section 8b inner contract: `rccl_allreduce_grads`): RCCL behind the
 * C ABI, bound with dlopen on first use (libomfs_splat.so does not link against librccl; a process that already holds one keeps
 * it).  One communicator per rank and process:
 *   rank 0: omfs_comm_unique_id(id) -> 128 bytes, handed to every rank by the host's own means (a file, MPI, a TCP store ...)
 *   every rank: omfs_comm_create(id, rank, world_size, &comm) (collective: returns when all ranks have joined; the calling thread's
 *               current HIP device is the rank's GPU), ... omfs_comm_destroy(comm).
 * The collectives are enqueued on `stream` and return (stream-ordered like every other entry point; fp32, sum):
 *   omfs_rccl_allreduce_grads : in place over `count` floats -- the [59][n_pad] gradient buffer, a plane range of it, or the
 *                               trainer's grad_store with the FLAME gradients in front ("full" / "compact" exchange)
 *   omfs_rccl_allgather       : all [world][count_per_rank] <- every rank's mine [count_per_rank] (the dL/dcolour planes of the
 *                               compact exchange; the parameter shards of the sharded one, in place when mine = all + rank * count)
 *   omfs_rccl_reduce_scatter  : shard [count_per_rank] <- this rank's part of the sum of full [world * count_per_rank]
 * The Python engine goes through torch.distributed by default (the same RCCL); OMFS_DP_IMPL=abi routes its "full" exchange here. */
int omfs_comm_unique_id(void* id_host128);
int omfs_comm_create(const void* id_host128, int rank, int world_size, void** comm_out);
int omfs_comm_destroy(void* comm);
int omfs_rccl_allreduce_grads(void* comm, float* grads, size_t count, void* stream);
int omfs_rccl_allgather(void* comm, const float* mine, float* all, size_t count_per_rank, void* stream);
int omfs_rccl_reduce_scatter(void* comm, const float* full, float* shard, size_t count_per_rank, void* stream);

/* (1-lambda) L1 + lambda (1-SSIM), 11x11 gaussian window, zero padding.
 * Writes dimage [3][H][W] and the scalar loss into loss_out[0].
 * scratch: 3 * 3*H*W + OMFS_LOSS_TAIL floats (ABI 6): three derivative maps, then one loss partial per 64 x 34 pixel strip
 * and channel (the backward launch adds them up: no reduction launch between the two passes); images of more than
 * OMFS_LOSS_TAIL strips are refused. */
#define OMFS_LOSS_TAIL 16384
int omfs_loss_l1_ssim(const float* image, const float* target, int width, int height, float lambda_dssim,
                      float* dimage, float* loss_out, float* scratch, void* stream);

/* Image ingress (ABI 6): a decoded 8-bit image src [src_height][src_width][channels] (channels 1, 3 or 4; a fourth channel
 * is the matte) and an optional separate matte mask [src_height][src_width] become the training target of a view: resized
 * to width x height by PIL's Image.BOX rule (equal-weight average of the source pixels whose centre falls into the output
 * pixel's footprint; identity without a resize), rounded to 8-bit levels, composited on bg_host[3] (value * m + (1 - m) *
 * bg) and written as planar fp32 out_f32 [3][height][width] in [0,1] and / or interleaved out_u8 [height][width][3] (either
 * may be NULL).  engine/train.py builds its targets with it, engine/render.py the gt/ images an evaluator compares with
 * (`validation_reporting.py:60-78`).                                                                                  */
int omfs_prepare_target(const uint8_t* src, int channels, int src_width, int src_height, const uint8_t* mask, int width,
                        int height, const float* bg_host, float* out_f32, uint8_t* out_u8, void* stream);

/* PNG egress on the device (ABI 6; SURVEY.md section 8f-3, `render_surgery.py:324-362` reads the frames back as PNG files):
 * the scanlines of omfs_image_to_png_rows (rows [height][1 + 3 width], filter type 0) become a COMPLETE zlib stream -- fixed
 * Huffman code, run-length matches one RGB pixel back, one deflate block per scanline closed by an empty stored block (byte
 * aligned: the rows' blocks concatenate bytewise), final empty block, Adler-32 of all scanlines.  The host wraps it into the
 * IDAT chunk (length, type, CRC-32) between IHDR and IEND; any PNG reader decodes it to the rows' pixels, bit for bit.
 *   slots  [height * omfs_png_slot_stride(width)] bytes, sizes [height], adler [2 * height] words: scratch;
 *   stream [stream_capacity >= height * omfs_png_slot_stride(width) + 16] bytes: the zlib stream;
 *   stream_len [1]: its length in bytes (0: capacity exceeded, which the sizes above rule out).                        */
int omfs_png_slot_stride(int width);
int omfs_png_deflate(const uint8_t* rows, int width, int height, uint8_t* slots, uint32_t* sizes, uint32_t* adler,
                     uint8_t* stream, uint32_t stream_capacity, uint32_t* stream_len, void* stream_hip);
/* Host-side fetch of such a frame -- the ONE entry point that blocks, so that an encoder thread of a Python host spends its
 * wait inside a single foreign call.  dev = [stream_len (4 bytes) | 12 bytes | stream ...] in one allocation, host_pinned the
 * same size; waits for ready_event (a hipEvent_t, may be NULL) on copy_stream (a stream of the calling thread), copies
 * 16 + guess_bytes speculatively and the remainder only if the stream is longer, synchronises copy_stream.  Returns the
 * stream length (the zlib stream is host_pinned + 16 ...), or a negative OMFS_ERR_*.                                   */
long long omfs_png_fetch(void* host_pinned, const void* dev, size_t capacity_bytes, size_t guess_bytes, void* copy_stream,
                         void* ready_event);

typedef struct omfs_adam_params {
  float lr[OMFS_NPLANES]; /* learning rate per plane                                              */
  float beta1, beta2, eps;
  int step;               /* 1-based                                                              */
  float grad_scale;       /* multiply gradients (1/world_size after a sum all-reduce)             */
} omfs_adam_params;

/* torch.optim.Adam semantics on params/m/v [59][n_pad] */
int omfs_adam_step(float* params, const float* grads, float* m, float* v, int n, int n_pad,
                   const omfs_adam_params* ap, void* stream);
/* the same step on planes [plane0, plane0 + n_planes) only (same ap->step for every part of one iteration): lets the
 * data-parallel trainer update the planes whose gradients are complete while the others are still being reduced */
int omfs_adam_step_planes(float* params, const float* grads, float* m, float* v, int n, int n_pad,
                          const omfs_adam_params* ap, int plane0, int n_planes, void* stream);
/* ABI 7: the same step on the 45 SH planes of degree >= 1 (planes 14..58) with their gradient formed where it is consumed:
 * plane 11 + 3k + c takes Y_k(dir) * drgb[c] for k < (sh_degree + 1)^2 and 0 beyond -- the very product omfs_project_bwd writes
 * into the gradient planes when gb->drgb_out is NULL (same operations in the same order: parameters and moments come out
 * bit-identical).  drgb, dir [3][n_pad] as omfs_project_bwd left them (gb->drgb_out, gb->dir_out).  Saves the 2 x 45 planes of
 * gradient traffic (write in project_bwd, read here) of a single-GPU training iteration.  ap->step as for the other parts.
 * grads_low (may be NULL): the gradient buffer; when given, planes 0..13 are updated from it in the SAME launch (what
 * omfs_adam_step_planes(.., 0, 14) does), so the whole Adam step of an iteration stays one launch.                        */
int omfs_adam_step_sh_rest(float* params, const float* grads_low, const float* drgb, const float* dir, float* m, float* v, int n,
                           int n_pad, const omfs_adam_params* ap, int sh_degree, void* stream);

/* The same step on the flat range [offset, offset + count) of the [59][n_pad] buffers (both multiples of 4): the shard a
 * data-parallel rank owns after a reduce-scatter of the gradient ("sharded" exchange: reduce-scatter, Adam on 1/W of the
 * elements, all-gather of the updated parameters).  params / grads / m / v point at the START of the range. */
int omfs_adam_step_range(float* params, const float* grads, float* m, float* v, int n_pad, long long offset, long long count,
                         const omfs_adam_params* ap, void* stream);

/* One training view from the projection to the parameter gradients in ONE call: omfs_project_fwd, the four binning
 * steps, omfs_composite_fwd, [omfs_rgb8_to_image when the target is 8-bit], omfs_loss_l1_ssim, omfs_composite_bwd,
 * omfs_project_bwd, enqueued in this order on `stream` exactly as the separate calls would be.  For hosts whose per-call
 * cost matters (a Python host pays ~10 us per ctypes call: at 5k Gaussians / 256x256 the device needs less per iteration
 * than twelve such calls take). */
typedef struct omfs_view_step {
  const omfs_gaussians* g;
  const float* face_xf;
  const omfs_camera* cam;
  const omfs_raster_buffers* rb;
  const omfs_grad_buffers* gb;
  const omfs_reg_params* reg;
  const float* target;         /* [3][H][W] fp32, or NULL when target_rgb8 is given                                     */
  const uint8_t* target_rgb8;  /* [H][W][3]; expanded into target_scratch [3][H][W] first                                */
  float* target_scratch;
  float lambda_dssim;
  float* loss_out;             /* [1]                                                                                    */
  float* loss_scratch;         /* 3*3*H*W + OMFS_LOSS_TAIL floats (omfs_loss_l1_ssim)                                      */
} omfs_view_step;
int omfs_view_forward_backward(const omfs_view_step* v, void* stream);
/* ABI 7: the same call in two halves, for a data-parallel host that issues a collective between them (SURVEY.md section 8e:
 * the all-gather of each rank's dL/dcolour runs under omfs_project_bwd).  The first half ends with omfs_composite_bwd and,
 * when drgb_out != NULL ([3][n_pad]), omfs_extract_drgb into it; the second half is omfs_project_bwd with the same struct.
 * omfs_view_forward_backward(v) == omfs_view_forward_composite_bwd(v, NULL) followed by omfs_view_project_bwd(v).     */
int omfs_view_forward_composite_bwd(const omfs_view_step* v, float* drgb_out, void* stream);
int omfs_view_project_bwd(const omfs_view_step* v, void* stream);

/* ---- device-resident per-iteration scalars.  With them nothing about a training iteration is a kernel ARGUMENT any more
 * (the learning-rate schedule and Adam's bias corrections were the last ones), so a whole iteration can be captured in a
 * hipGraph once per view and replayed.  omfs_step_advance (one thread) increments both step counters and derives, in double
 * precision like the host-side entry points: the position learning rate lr_xyz = exp(log(lr_init)(1-t) + log(lr_final) t),
 * t = min(iterations completed / max_steps, 1) (3DGS' log-linear schedule), and Adam's bias corrections for the
 * Gaussian and the FLAME parameters.  omfs_adam_step_dev / omfs_adam_flat_multi(state_dev != NULL) read them: planes 0..2
 * take lr_xyz, every other plane ap->lr[plane]; ap->step / step are ignored.                                            */
typedef struct omfs_step_state {          /* 16 words of DEVICE memory; the caller initialises step / flame_step        */
  int32_t step;                           /* Adam step of the Gaussian parameters (1-based after omfs_step_advance)      */
  int32_t flame_step;                     /* Adam step of the FLAME parameters                                           */
  float lr_xyz;                           /* position learning rate of this iteration                                    */
  float inv_bc1, inv_sqrt_bc2;            /* 1 / (1 - beta1^step), 1 / sqrt(1 - beta2^step)                              */
  float flame_inv_bc1, flame_inv_sqrt_bc2;
  int32_t table_base;                     /* iteration (0-based) that entry 0 of omfs_step_advance's next_table belongs to  */
  float reserved[8];
} omfs_step_state;
typedef struct omfs_lr_schedule {
  float lr_init, lr_final;
  int max_steps;
  float beta1, beta2;
} omfs_lr_schedule;
/* next_table (device, may be NULL): entry j holds a per-iteration word of iteration table_base + j (the trainer stores the
 * FLAME timestep of that iteration's view); the entry of the iteration AFTER the one being advanced to is copied to
 * next_out[0] -- a captured iteration poses the next view's FLAME frame from it, so a replay needs no host-side update
 * at all (index clamped to the table). */
int omfs_step_advance(omfs_step_state* state_dev, const omfs_lr_schedule* sch, const int32_t* next_table, int table_len,
                      int32_t* next_out, void* stream);
int omfs_adam_step_dev(float* params, const float* grads, float* m, float* v, int n, int n_pad, const omfs_adam_params* ap,
                       const omfs_step_state* state_dev, int plane0, int n_planes, void* stream);

/* ---- adaptive density control (clone / split / prune), SURVEY.md Appendix A item 10: what the absent upstream train.py does
 * between iterations (call site 02_Visual_Engine/train_ghost.py:227-271).  Three steps, the caller allocates:
 *   classify : cls[i] = 1 (kept) | 2 (cloned) | 4 (split) from the accumulated statistics (stats [2][n_pad]: sum of the
 *              view-space positional gradient norms, number of views that saw the Gaussian), the largest world-space axis
 *              exp(max log-scale) * triangle scale (face_xf [F][16], word 12) and the opacity:
 *                hot = mean gradient >= grad_threshold;  clone = hot && size <= size_threshold;  split = hot && !clone;
 *                pruned = split || sigmoid(opacity) < min_opacity || (prune_size > 0 && size > prune_size);
 *              block_counts [3][ceil(n/256)] receives the per-256 counts of the three bits; grad_out [n] (optional) the
 *              mean gradients
 *   scan     : block_counts -> exclusive offsets in place, totals[3] = number kept / cloned / split
 *   compact  : writes [kept | clones | first children | second children], each group in index order, into params_out
 *              [59][n_out_pad], binding_out, m_out, v_out (all zero-filled by the caller, n_out = totals[0] + totals[1]
 *              + 2 totals[2] <= n_out_pad); kept Gaussians carry their Adam moments, new ones start from zero; the children
 *              of a split are two samples of the Gaussian in its local frame (counter-based generator keyed by seed_lo,
 *              seed_hi, index, child, axis -- identical on every rank) with log-scales lowered by log 1.6          */
typedef struct {
  float grad_threshold, size_threshold, min_opacity, prune_size;
  uint32_t seed_lo, seed_hi;
} omfs_densify_params;
int omfs_densify_classify(const omfs_gaussians* g, const float* face_xf, const float* stats, const omfs_densify_params* p,
                          uint8_t* cls, float* grad_out, uint32_t* block_counts, void* stream);
int omfs_densify_scan(uint32_t* block_counts, int n, uint32_t* totals, void* stream);
int omfs_densify_compact(const omfs_gaussians* g, const float* adam_m, const float* adam_v, const uint8_t* cls,
                         const uint32_t* block_offsets, const uint32_t* totals, const omfs_densify_params* p, int n_out_pad,
                         float* params_out, int32_t* binding_out, float* m_out, float* v_out, void* stream);

/* #Gaussians with radius>0 of the last projected view -> count_out[0] (device) */
int omfs_count_visible(const omfs_raster_buffers* rb, int n, uint32_t* count_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OMFS_SPLAT_H */
