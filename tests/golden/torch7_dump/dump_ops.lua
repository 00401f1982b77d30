-- Out-of-band pin of every third-party ([3P]) op the oracle restates, plus one Torch-written model file (DESIGN.md s5,
-- VERDICT r03 next #5).  Run ONCE on a machine with Torch7 + cutorch/cunn/cudnn + the reference checkout (stn, spy built),
-- from the root of the reference repository:
--     th /path/to/this/dump_ops.lua /path/to/out_dir
-- and drop out_dir/* into tests/golden/torch7_dump/ops/ of this repository.  tests/test_reference_dump.py regenerates
-- the inputs (same counter generator, below), runs the CPU oracle (and, on a GPU box, the library) and compares.
-- No pretrained weights are needed for this script.  Files: <name>.f32 = raw little-endian float32, C order;
-- ops_meta.txt = one line per file "<name> <ndim> <d1> ... <dn>"; tiny_model.t7 = torch.save of createModelMulti().
require 'torch'
require 'nn'
require 'cutorch'
require 'cunn'
require 'cudnn'
require 'nngraph'
require 'image'
require 'stn'
require 'spy'
local out = arg[1] or '.'
paths.mkdir(out)
torch.setdefaulttensortype('torch.FloatTensor')

-- inputs: 32-bit multiplicative counter generator, exact in double arithmetic on both sides
--   s <- (s * 69069 + 1) mod 2^32 ;  value = s / 2^32 - 0.5          (tests/test_reference_dump.py: lcg())
local function lcg(seed, ...)
   local t = torch.FloatTensor(...)
   local flat = t:view(-1)
   local s = seed
   for i = 1, flat:size(1) do
      s = (s * 69069 + 1) % 4294967296
      flat[i] = s / 4294967296 - 0.5
   end
   return t
end

local meta = io.open(out .. '/ops_meta.txt', 'w')
local function dump(name, tensor)
   local t = tensor:float():contiguous()
   local f = torch.DiskFile(out .. '/' .. name .. '.f32', 'w'):binary():littleEndianEncoding()
   f:writeFloat(t:storage())
   f:close()
   local line = name .. ' ' .. t:dim()
   for d = 1, t:dim() do line = line .. ' ' .. t:size(d) end
   meta:write(line .. '\n')
end

-- 1, 2: nn.CostVolMulti(9, true / false) on {ref, frm}, B x C x h x w (models/CostVolMulti.lua:48-109)
paths.dofile('models/CostVolMulti.lua')
do
   local ref = lcg(1001, 2, 16, 12, 20):mul(2):cuda()
   local frm = lcg(1002, 2, 16, 12, 20):mul(2):cuda()
   dump('costvol_fwd', nn.CostVolMulti(9, true):cuda():forward({ref, frm}))
   dump('costvol_bwd', nn.CostVolMulti(9, false):cuda():forward({ref, frm}))
end

-- 3: the warping unit of models/pwc.lua:68-73 (CUDA nn.BilinearSamplerBHWD of extras/stnbhwd on a pixel-flow grid);
--    flows of up to +-6 pixels: the border clamp is exercised along every edge
do
   local img = lcg(3001, 2, 8, 12, 20):cuda()
   local flow = lcg(3002, 2, 2, 12, 20):mul(12):cuda()
   local I = nn.Identity()()
   local F = nn.Identity()()
   local input = I - nn.Transpose({2,3}, {3,4})
   local fl = F - nn.Transpose({2,3}, {3,4})
   local W = {input, fl} - nn.BilinearSamplerBHWD() - nn.Transpose({3,4}, {2,3})
   local unit = nn.gModule({I, F}, {W}):cuda()
   dump('warp', unit:forward({img, flow}))
end

-- 4, 5, 6: the stock nn modules the graph uses (models/pwc.lua:308-319, 360-380)
do
   local x = lcg(4001, 2, 2, 6, 10):cuda()
   dump('upsample_bilinear2', nn.SpatialUpSamplingBilinear(2.0):cuda():forward(x))
   dump('upsample_nearest2', nn.SpatialUpSamplingNearest(2.0):cuda():forward(x))
   local l = lcg(5001, 2, 2, 6, 10):mul(8):cuda()
   dump('softmax', nn.SpatialSoftMax(true):cuda():forward(l))
end

-- 7: image.scale as computeFlow uses it (back2future.lua:71,80): bilinear 375 x 1242 -> 320 x 1216, 'simple' back
do
   local im = image.load('samples/frame_0009.png'):float()
   local down = image.scale(im, 1216, 320)
   dump('scale_bilinear', down)
   dump('scale_simple', image.scale(down[{{1, 2}}]:double(), im:size(3), im:size(2), 'simple'))
end

-- 8: a Torch-written model file + its whole output table: createModelMulti() with its defaults (models/pwc.lua:88-100:
--    win 5, 4 levels, skip 2, random weights) on a counter-generated 1 x 9 x 64 x 128 input
do
   paths.dofile('util.lua')
   paths.dofile('models/pwc.lua')
   torch.manualSeed(7)
   cutorch.manualSeed(7)
   local model = createModelMulti()
   if torch.type(model) == 'nn.DataParallelTable' then model = model:get(1) end
   model:evaluate()
   local x = lcg(8001, 1, 9, 64, 128):cuda()
   local est = model:forward(x)
   for i = 1, #est do dump(string.format('est_%02d', i), est[i]) end
   model:clearState()
   torch.save(out .. '/tiny_model.t7', model)
end
meta:close()
print('wrote ' .. out)
