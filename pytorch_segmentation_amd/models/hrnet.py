"""HRNet (W32-style four-resolution network) on the HIP kernels.

Constructor, attribute names (`stem`, `transition1..3`, `stage2..4`, `final_layer`; `branches` / `fuse_layers` inside a
module) and the nesting of Sequential / ModuleList containers follow the reference's models/hrnet.py:27-406, so
state-dicts interchange.  Arithmetic:

  stem          ConvNormAct(3,64,3,s2,act=None) -> ConvNormAct(64,64,3,s2) -> 4 Bottlenecks (64 -> 256)   (:235-237)
  transition_k  a new, half-resolution branch = ConvNormAct(3,s2) of the LAST branch of the previous stage (:282-303,
                :374-398); existing branches pass through
  stage_k       one HRModule: 4 BasicBlocks per branch, then out_i = relu(sum_j f_ij(x_j)) with
                f_ij = identity (j == i) | bilinear(x 2^(j-i), align_corners=False) o ConvNormAct 1x1 (j > i)
                     | a chain of ConvNormAct 3x3 s2, the last without activation (j < i)              (:185-229)
  head          final_layer 1x1 (bias) on the full-resolution branch, then x4 bilinear align_corners=False (:400-404)

On the GPU the sum over j never materialises its terms separately: a down-sampling chain's last BatchNorm pass adds
the running sum as its residual operand, an up-sampled term is added in place, and the ReLU rides on the last term.
In backward the ReLU mask is applied once per output and every term consumes that one masked gradient; gradients of
a branch output that is used by several outputs merge in the dgrad epilogues (`dx_accumulate`).
"""
import torch.nn as nn

from .. import ops
from ..backbones.resnet import Bottleneck
from ..nn import ACT_NONE, ACT_RELU, BatchNorm2d, Conv2d, ConvNormAct, initialize_weights, loss_grad_in
from ..ops import Act

BN_MOMENTUM = 0.1


class BasicBlock(nn.Module):
    """relu(bn2(conv2(relu(bn1(conv1(x))))) + residual) -- reference models/hrnet.py:27-56."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.downsample = downsample
        self.stride = stride

    def fwd(self, x, env):
        y1, st1, s1 = self.conv1.fwd(x, env, want_stats=self.bn1.training)
        z1, b1 = self.bn1.fwd(y1, st1, env, act=ACT_RELU)
        y2, st2, s2 = self.conv2.fwd(z1, env, want_stats=self.bn2.training)
        sd = bd = None
        identity = x
        if self.downsample is not None:
            dconv, dbn = self.downsample[0], self.downsample[1]
            yd, std, sd = dconv.fwd(x, env, want_stats=dbn.training)
            identity, bd = dbn.fwd(yd, std, env, act=ACT_NONE)
        out, b2 = self.bn2.fwd(y2, st2, env, act=ACT_RELU, residual=identity)
        return out, (s1, b1, s2, b2, sd, bd)

    def bwd(self, dout, saved, env):
        s1, b1, s2, b2, sd, bd = saved
        d_id = dout.like()
        dy2 = self.bn2.bwd(dout, b2, env, dres=d_id)
        dz1 = self.conv2.bwd(dy2, s2, env)
        dy1 = self.bn1.bwd(dz1, b1, env)
        if self.downsample is not None:
            dconv, dbn = self.downsample[0], self.downsample[1]
            dyd = dbn.bwd(d_id, bd, env)
            dx = dconv.bwd(dyd, sd, env)
            self.conv1.bwd(dy1, s1, env, dx_out=dx, dx_accumulate=True)
            return dx
        self.conv1.bwd(dy1, s1, env, dx_out=d_id, dx_accumulate=True)
        return d_id


def _residual_downsample(inplanes, outplanes, stride):
    return nn.Sequential(Conv2d(inplanes, outplanes, 1, stride=stride, bias=False),
                         BatchNorm2d(outplanes, momentum=BN_MOMENTUM))


def _chain(seq):
    """The ConvNormAct blocks of a down-sampling fuse / transition entry (reference nests the non-final ones in an
    extra nn.Sequential, models/hrnet.py:216-220)."""
    return [m if isinstance(m, ConvNormAct) else m[0] for m in seq]


class HRModule(nn.Module):
    def __init__(self, num_branches, blocks, num_blocks, num_inchannels, num_channels, multi_scale_output=True):
        super().__init__()
        for name, lst in (('NUM_BLOCKS', num_blocks), ('NUM_CHANNELS', num_channels), ('NUM_INCHANNELS', num_inchannels)):
            if num_branches != len(lst):
                raise ValueError('NUM_BRANCHES({}) <> {}({})'.format(num_branches, name, len(lst)))
        self.num_inchannels = num_inchannels
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        self.branches = nn.ModuleList([self._make_one_branch(i, blocks, num_blocks, num_channels)
                                       for i in range(num_branches)])
        self.fuse_layers = self._make_fuse_layers()
        self.relu = nn.ReLU(True)
        initialize_weights(self)

    def _make_one_branch(self, index, block, num_blocks, num_channels, stride=1):
        downsample = None
        if stride != 1 or self.num_inchannels[index] != num_channels[index] * block.expansion:
            downsample = _residual_downsample(self.num_inchannels[index], num_channels[index] * block.expansion, stride)
        layers = [block(self.num_inchannels[index], num_channels[index], stride, downsample)]
        self.num_inchannels[index] = num_channels[index] * block.expansion
        for _ in range(1, num_blocks[index]):
            layers.append(block(self.num_inchannels[index], num_channels[index]))
        return nn.Sequential(*layers)

    def _make_fuse_layers(self):
        if self.num_branches == 1:
            return None
        n, ch = self.num_branches, self.num_inchannels
        fuse_layers = []
        for i in range(n if self.multi_scale_output else 1):
            row = []
            for j in range(n):
                if j > i:
                    row.append(nn.Sequential(ConvNormAct(ch[j], ch[i], 1),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode='bilinear',
                                                         align_corners=False)))
                elif j == i:
                    row.append(None)
                else:
                    convs = []
                    for k in range(i - j):
                        if k == i - j - 1:
                            convs.append(ConvNormAct(ch[j], ch[i], 3, 2, activate=None))
                        else:
                            convs.append(nn.Sequential(ConvNormAct(ch[j], ch[j], 3, 2)))
                    row.append(nn.Sequential(*convs))
            fuse_layers.append(nn.ModuleList(row))
        return nn.ModuleList(fuse_layers)

    def get_num_inchannels(self):
        return self.num_inchannels

    # ---- explicit forward / backward over lists of branch activations
    def fwd(self, xs, env):
        n = self.num_branches
        xs, saved_br = list(xs), [None] * n
        dev = xs[0].device
        # the branches are independent chains: a lane each while a step is being captured (ops.Branches)
        br = ops.Branches(dev, n)
        for i in range(n):
            with br.lane(i, xs[i]):
                sl, cur = [], xs[i]
                for blk in self.branches[i]:
                    cur, sb = blk.fwd(cur, env)
                    sl.append(sb)
                xs[i] = cur
                saved_br[i] = sl
        br.join(*xs)
        if n == 1:
            return [xs[0]], (saved_br, None, None)
        # ... and so are the fused outputs: row i reads every branch and accumulates its own sum
        outs, saved_fuse = [], []
        br = ops.Branches(dev, len(self.fuse_layers))
        for i, row in enumerate(self.fuse_layers):
            with br.lane(i, *xs):
                acc, sf = None, []
                for j in range(n):
                    act = ACT_RELU if j == n - 1 else ACT_NONE   # the module's ReLU rides on the last term of the sum
                    if j == i:
                        if acc is None:
                            acc = xs[j]                          # read-only alias; the next term writes a fresh buffer
                        else:
                            z = xs[j].like()
                            ops.bn_act_fwd(xs[j], None, act, z, residual=acc)
                            acc = z
                        sf.append(None)
                    elif j > i:
                        t, s = row[j][0].fwd(xs[j], env)
                        u = t.new(t.B, xs[i].H, xs[i].W, t.C)
                        ops.bilinear_fwd(t, u, False)
                        ops.bn_act_fwd(u, None, act, u, residual=acc)      # u = act(u + acc), in place
                        acc = u
                        sf.append((s, (t.B, t.H, t.W, t.C)))
                    else:
                        chain, cur, sc = _chain(row[j]), xs[j], []
                        for cna in chain[:-1]:
                            cur, s = cna.fwd(cur, env)
                            sc.append(s)
                        last = chain[-1]
                        y, st, s_c = last.conv.fwd(cur, env, want_stats=last.bn.training)
                        acc, s_b = last.bn.fwd(y, st, env, act=ACT_NONE, residual=acc)   # BN(y) + running sum, one pass
                        sc.append((s_c, s_b))
                        sf.append(sc)
                outs.append(acc)
                saved_fuse.append(sf)
        br.join(*outs)
        return outs, (saved_br, saved_fuse, outs if env.save else None)

    def bwd(self, douts, saved, env):
        """douts: gradients of the returned outputs (entries may be None).  Returns the per-branch input gradients.

        Organised by COLUMN: lane j gathers every term that read branch j's output -- in row order, the order in which the
        single-stream code of rounds 1-3 added them -- and then walks branch j's blocks backwards; the lanes meet again
        only at the end of the module."""
        saved_br, saved_fuse, outs = saved
        n = self.num_branches
        dev = next(d for d in douts if d is not None).device
        if n == 1:
            d = douts[0]
            for blk, sb in zip(reversed(list(self.branches[0])), reversed(saved_br[0])):
                d = blk.bwd(d, sb, env)
            return [d]
        rows = [i for i in range(len(self.fuse_layers)) if douts[i] is not None]
        dss = {}
        for i in rows:
            ds = outs[i].like()
            ops.act_bwd(douts[i], outs[i], ACT_RELU, ds)               # gradient of the pre-ReLU sum, shared by all terms
            dss[i] = ds
        dxs = [None] * n
        br = ops.Branches(dev, n)
        for j in range(n):
            with br.lane(j, *dss.values()):
                dx = None
                for i in rows:
                    ds, sf, row = dss[i], saved_fuse[i], self.fuse_layers[i]
                    if j > i:
                        s, tshape = sf[j]
                        dt = ds.new(*tshape)
                        ops.bilinear_bwd(ds, dt, False)
                        dx = row[j][0].bwd(dt, s, env, dx_out=dx, dx_accumulate=dx is not None)
                    elif j < i:
                        chain, sc = _chain(row[j]), sf[j]
                        s_c, s_b = sc[-1]
                        d = chain[-1].bn.bwd(ds, s_b, env)
                        for k in range(len(chain) - 1, -1, -1):
                            tgt = dx if k == 0 else None
                            if k == len(chain) - 1:
                                d = chain[k].conv.bwd(d, s_c, env, dx_out=tgt, dx_accumulate=tgt is not None)
                            else:
                                d = chain[k].bwd(d, sc[k], env, dx_out=tgt, dx_accumulate=tgt is not None)
                        dx = d
                    elif dx is None:
                        # identity term first in its column: a COPY becomes the accumulator -- the other columns read ds too, on
                        # other lanes, so ds is never handed on as this column's gradient (not even where nothing is added to
                        # it afterwards: a block whose backward wrote into its incoming gradient would race with those readers
                        # in a replay only; one copy per module buys not having that contract -- ADVICE r4)
                        dx = ds.like()
                        ops.copy2d(ds, dx)
                    else:
                        ops.copy2d(ds, dx, accumulate=True)
                if dx is not None:
                    for blk, sb in zip(reversed(list(self.branches[j])), reversed(saved_br[j])):
                        dx = blk.bwd(dx, sb, env)
                dxs[j] = dx
        br.join(*dxs)
        return dxs


class HRNet(nn.Module):
    # a captured step deals the weight gradients onto TWO lanes: here (104 convs, most of them narrow) the one stream they
    # would share is the longest lane of the replayed backward pass (ops.fork_aux, utils/trainer.py)
    capture_wgrad_lanes = 2
    # ... and the step lives on the replay (1000 launches): main chain + two weight-gradient lanes + two branch lanes; the Trainer
    # reserves the lane executor's stream pool for them up front
    replay_lanes = 5

    def __init__(self, num_classes=2, num_branches_list=[2, 3, 4]):
        super().__init__()
        self.inplanes = 64
        block = BasicBlock
        self.stem = nn.Sequential(ConvNormAct(3, 64, 3, 2, activate=None), ConvNormAct(64, 64, 3, 2),
                                  self._make_layer(Bottleneck, 64, 4))
        pre = [256]
        for k, nb in enumerate(num_branches_list):
            num_channels = [32 * (2 ** i) for i in range(nb)]
            num_inchannels = [c * block.expansion for c in num_channels]
            setattr(self, 'transition%d' % (k + 1), self._make_transition_layer(pre, num_channels))
            stage, pre = self._make_stage(nb, [4] * nb, num_channels, block, num_inchannels,
                                          multi_scale_output=(k != len(num_branches_list) - 1))
            setattr(self, 'stage%d' % (k + 2), stage)
        self.num_stages = len(num_branches_list)
        self.final_layer = Conv2d(pre[0], num_classes, 1)
        self.num_classes = num_classes

    def _make_transition_layer(self, pre, cur):
        layers = []
        for i in range(len(cur)):
            if i < len(pre):
                layers.append(ConvNormAct(pre[i], cur[i], 3) if cur[i] != pre[i] else None)
            else:
                convs = []
                for j in range(i + 1 - len(pre)):
                    cout = cur[i] if j == i - len(pre) else pre[-1]
                    convs.append(ConvNormAct(pre[-1], cout, 3, 2))
                layers.append(nn.Sequential(*convs))
        return nn.ModuleList(layers)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = _residual_downsample(self.inplanes, planes * block.expansion, stride)
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def _make_stage(self, num_branches, num_blocks, num_channels, block, num_inchannels, multi_scale_output=True,
                    num_modules=1):
        modules = []
        for i in range(num_modules):
            mso = multi_scale_output or i != num_modules - 1      # only the last module may drop the extra outputs
            modules.append(HRModule(num_branches, block, num_blocks, num_inchannels, num_channels, mso))
            num_inchannels = modules[-1].get_num_inchannels()
        return nn.Sequential(*modules), num_inchannels

    # ---- transitions: entry i is None (pass-through) or a ConvNormAct / chain applied to the LAST previous branch
    @staticmethod
    def _trans_fwd(trans, ys, env):
        xs, saved = [], []
        for i, t in enumerate(trans):
            if t is None:
                xs.append(ys[i])
                saved.append(None)
                continue
            cur, sc = ys[-1], []
            for cna in _chain([t] if isinstance(t, ConvNormAct) else t):
                cur, s = cna.fwd(cur, env)
                sc.append(s)
            xs.append(cur)
            saved.append(sc)
        return xs, saved

    @staticmethod
    def _trans_bwd(trans, dxs, saved, npre, env):
        dys = [None] * npre
        for i, t in enumerate(trans):
            if t is None:
                dys[i] = dxs[i]
        for i, t in enumerate(trans):
            if t is None or dxs[i] is None:
                continue
            chain, d = _chain([t] if isinstance(t, ConvNormAct) else t), dxs[i]
            for k in range(len(chain) - 1, -1, -1):
                tgt = dys[npre - 1] if k == 0 else None
                d = chain[k].bwd(d, saved[i][k], env, dx_out=tgt, dx_accumulate=tgt is not None)
            dys[npre - 1] = d
        return dys

    def model_fwd(self, x, env):
        xa = Act.from_nchw(x, 8 if env.half else 4, dtype=env.act_dtype)
        z0, s0 = self.stem[0].fwd(xa, env)
        cur, s1 = self.stem[1].fwd(z0, env)
        s_layer = []
        for blk in self.stem[2]:
            cur, sb = blk.fwd(cur, env)
            s_layer.append(sb)
        ys, s_stages = [cur], []
        for k in range(self.num_stages):
            trans, stage = getattr(self, 'transition%d' % (k + 1)), getattr(self, 'stage%d' % (k + 2))
            npre = len(ys)
            xs, s_t = self._trans_fwd(trans, ys, env)
            s_mods = []
            for mod in stage:
                xs, sm = mod.fwd(xs, env)
                s_mods.append(sm)
            ys = xs
            s_stages.append((s_t, s_mods, npre))
        lr, _, s_fin = self.final_layer.fwd(ys[0], env, out_f32=True)     # (half policy: the logits leave in fp32)
        out = ops.bilinear_fwd_nchw(lr, self.num_classes, lr.H * 4, lr.W * 4, False)
        return out, (s0, s1, s_layer, s_stages, s_fin, (lr.B, lr.H, lr.W, lr.C), len(ys))

    def model_bwd(self, dout, saved, env):
        s0, s1, s_layer, s_stages, s_fin, lshape, nout = saved
        dlr = Act.empty(*lshape, dout.device, zero=True)             # padded class channels stay zero
        ops.bilinear_bwd_nchw(dout, dlr, self.num_classes, False)
        dys = [self.final_layer.bwd(loss_grad_in(dlr, env), s_fin, env)] + [None] * (nout - 1)
        for k in range(self.num_stages - 1, -1, -1):
            trans, stage = getattr(self, 'transition%d' % (k + 1)), getattr(self, 'stage%d' % (k + 2))
            s_t, s_mods, npre = s_stages[k]
            for mod, sm in zip(reversed(list(stage)), reversed(s_mods)):
                dys = mod.bwd(dys, sm, env)
            dys = self._trans_bwd(trans, dys, s_t, npre, env)
        d = dys[0]
        for blk, sb in zip(reversed(list(self.stem[2])), reversed(s_layer)):
            d = blk.bwd(d, sb, env)
        d = self.stem[1].bwd(d, s1, env)
        self.stem[0].bwd(d, s0, env, need_dx=False)                  # image gradient is never needed

    def forward(self, x):
        from ..bridge import run_model
        return run_model(self, x)
