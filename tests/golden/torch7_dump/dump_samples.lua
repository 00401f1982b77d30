-- Out-of-band pin for the parity-unpinned oracle (DESIGN.md s5): run ONCE on a machine that has Torch7 + the
-- reference checkout + the pretrained weights, from the root of the reference repository:
--     th /path/to/this/dump_samples.lua Ours-Soft-ft-KITTI /path/to/out_dir
-- and drop the produced files (plus a copy of the .t7 the name maps to, as weights.t7) into
-- tests/golden/torch7_dump/ of this repository.  tests/test_reference_dump.py then compares the CPU oracle and the
-- GPU library against them (max-abs <= 1e-3 on the flow; masks equal away from the 0.6666 threshold).
-- Raw little-endian dumps, no Torch serialization: flow.f64 (2 x H x W doubles), fwd_occ.u8 / bwd_occ.u8 (H x W
-- bytes), meta.txt ("name H W").
local name = arg[1] or 'Ours-Soft-ft-KITTI'
local out = arg[2] or '.'
local image = require 'image'
local back2future = require('back2future')
local computeFlow = back2future.init(name)
local im1 = image.load('samples/frame_0009.png')
local im2 = image.load('samples/frame_0010.png')
local im3 = image.load('samples/frame_0011.png')
local flow, fwd_occ, bwd_occ = computeFlow(im1, im2, im3)
local function dump(path, tensor)
   local f = torch.DiskFile(path, 'w'):binary():littleEndianEncoding()
   local s = tensor:contiguous():storage()
   if tensor:type() == 'torch.DoubleTensor' then f:writeDouble(s) else f:writeByte(s) end
   f:close()
end
dump(out .. '/flow.f64', flow:double())
dump(out .. '/fwd_occ.u8', fwd_occ:byte())
dump(out .. '/bwd_occ.u8', bwd_occ:byte())
local m = io.open(out .. '/meta.txt', 'w')
m:write(string.format('%s %d %d\n', name, flow:size(2), flow:size(3)))
m:close()
