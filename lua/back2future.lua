----------------------------------------------------------------------------
-- back2future.lua -- drop-in replacement of the reference's inference module
-- (/root/reference/back2future.lua) on top of libb2f.so (MI355X / gfx950).
--
--   back2future = require('back2future')
--   computeFlow = back2future.init('Ours-Soft-ft-KITTI')
--   flow, fwd_occ, bwd_occ = computeFlow(im1, im2, im3)
--
-- Same module table (init, normalize), same argument and return order and types
-- as back2future.lua:45-130: im* are 3xHxW torch tensors in [0,1] (image.load),
-- flow is a 2xHxW torch.DoubleTensor, the masks are 1xHxW torch.ByteTensor.
-- Needs LuaJIT (ffi) and torch7's CPU tensors only: no cutorch / cunn / cudnn /
-- nngraph / stn / spy.  The model directory convention ('models/RoamingImages_*.t7'
-- relative to the current directory, back2future.lua:100-110) is implemented inside
-- b2f_init.
--
-- NOTE: this build image has no LuaJIT / Torch7, so this file is shipped UNTESTED;
-- back2future_amd/back2future.py is the tested mirror of the same calls
-- (INTEGRATION.md).
----------------------------------------------------------------------------
local ffi = require 'ffi'
require 'torch'

ffi.cdef[[
typedef struct b2f_ctx b2f_ctx;
const char *b2f_last_error(void);
int  b2f_init(const char *name_or_path, int device, b2f_ctx **out);
void b2f_destroy(b2f_ctx *ctx);
int  b2f_info(const b2f_ctx *ctx, int *levels, int *win, int *past_flow, int *n_outputs, long long *n_params);
int  b2f_compute_flow(b2f_ctx *ctx, const float *im1, const float *im2, const float *im3,
                      int H0, int W0, double *flow, unsigned char *fwd_occ, unsigned char *bwd_occ);
typedef struct b2f_multi b2f_multi;
int  b2f_init_multi(const char *name_or_path, int n_gpus, const int *devices, b2f_multi **out);
void b2f_destroy_multi(b2f_multi *m);
int  b2f_multi_compute_flow_batch(b2f_multi *m, int n, const float *im1, const float *im2, const float *im3,
                                  int H0, int W0, double *flow, unsigned char *fwd_occ, unsigned char *bwd_occ);
]]

local lib = ffi.load(os.getenv('B2F_LIB') or 'libb2f.so')

local M = {}

local meanstd = {
   mean = { 0.485, 0.456, 0.406 },
   std = { 0.229, 0.224, 0.225 },
}

local function check(rc)
   if rc ~= 0 then error(ffi.string(lib.b2f_last_error())) end   -- THError -> Lua error, as in the reference
end

-- M.normalize = TF.ColorNormalize(meanstd)  (back2future.lua:42-45, transforms.lua:33-45)
local function normalize(imgs)
   local img = imgs:clone()
   local chs = (img:size(1) / 3) - 1
   for c = 0, chs do
      for i = 1, 3 do
         img[3 * c + i]:add(-meanstd.mean[i])
         img[3 * c + i]:div(meanstd.std[i])
      end
   end
   return img
end
M.normalize = normalize

local function init(opt)
   opt = opt or 'Ours-Soft-ft-KITTI'
   local pctx = ffi.new('b2f_ctx*[1]')
   check(lib.b2f_init(opt, tonumber(os.getenv('B2F_DEVICE') or 0), pctx))
   local ctx = ffi.gc(pctx[0], lib.b2f_destroy)

   -- `channels` is a global in the reference (back2future.lua:120-126); kept for scripts that read it
   channels = 9

   local computeFlow = function(im1, im2, im3)
      local a, b, c = im1:float():contiguous(), im2:float():contiguous(), im3:float():contiguous()
      assert(a:dim() == 3 and a:size(1) == 3, 'expected 3 x H x W images')
      assert(a:isSameSizeAs(b) and a:isSameSizeAs(c), 'the three frames must have the same size')
      local height, width = a:size(2), a:size(3)
      print(width - math.fmod(width, 64), height - math.fmod(height, 64))   -- back2future.lua:69
      local flow_est = torch.DoubleTensor(2, height, width)
      local fwd_occ_est = torch.ByteTensor(1, height, width)
      local bwd_occ_est = torch.ByteTensor(1, height, width)
      check(lib.b2f_compute_flow(ctx, a:data(), b:data(), c:data(), height, width,
                                 flow_est:data(), fwd_occ_est:data(), bwd_occ_est:data()))
      return flow_est, fwd_occ_est, bwd_occ_est
   end
   return computeFlow
end
M.init = init

-- Several GPUs of one node (replaces nn.DataParallelTable, util.lua:27-48): initMulti(opt, nGPU) returns
-- computeFlowBatch(im1, im2, im3) on n x 3 x H x W tensors; the n triplets are split over the GPUs by the
-- library, the weights are broadcast to every GPU once (RCCL).  nGPU = 0: all visible GPUs.
local function initMulti(opt, nGPU)
   opt = opt or 'Ours-Soft-ft-KITTI'
   local pm = ffi.new('b2f_multi*[1]')
   check(lib.b2f_init_multi(opt, nGPU or 0, nil, pm))
   local m = ffi.gc(pm[0], lib.b2f_destroy_multi)
   return function(im1, im2, im3)
      local a, b, c = im1:float():contiguous(), im2:float():contiguous(), im3:float():contiguous()
      assert(a:dim() == 4 and a:size(2) == 3, 'expected n x 3 x H x W batches')
      assert(a:isSameSizeAs(b) and a:isSameSizeAs(c), 'the three frame batches must have the same size')
      local n, height, width = a:size(1), a:size(3), a:size(4)
      local flow_est = torch.DoubleTensor(n, 2, height, width)
      local fwd_occ_est = torch.ByteTensor(n, 1, height, width)
      local bwd_occ_est = torch.ByteTensor(n, 1, height, width)
      check(lib.b2f_multi_compute_flow_batch(m, n, a:data(), b:data(), c:data(), height, width,
                                             flow_est:data(), fwd_occ_est:data(), bwd_occ_est:data()))
      return flow_est, fwd_occ_est, bwd_occ_est
   end
end
M.initMulti = initMulti

return M
