"""Searched decoders: MicroDecoder (CVPR'19 cells) and TemplateDecoder (WACV'20
templates), built from a genotype list on top of the HIP-backed op registry.

Mirrors src/nn/micro_decoders.py: constructor / forward contracts, module tree
(state_dict names ``adapt{n}``, ``cells.{i}.op_{1,2}._ops.{j}...``, ``cells.{i}.agg``,
``aux_clfs.{i}.aux_clf``, ``_ops.{block}.{i}...``, ``pre_clf``, ``conv_clf``),
attributes read by callers (``info``, ``num_classes``, ``collect_inds`` /
``_collect_inds``, ``prettify``) and the same genotype semantics:

  MicroDecoder   config = [[op0, [p1, p2, o1, o2] x (L-1)], [[i, j] x cells]]
  TemplateDecoder config = [[op1, op2, agg] x T, [[pos1, pos2, cell, rep, stride_log2] x L]]

Indices refer to rl.genotypes.OP_NAMES / OP_NAMES_WACV / AGG_OP_NAMES.
"""
import torch.nn as nn

from .. import functional as F
from ..rl.genotypes import AGG_OP_NAMES, OP_NAMES, OP_NAMES_WACV
from .layer_factory import AGG_OPS, OPS, conv3x3, conv_bn_relu, run_op
from .modules import TREE_VERSION, FusedSequential


def _takes_pending_types():
    from .layer_factory import DilConv, SepConv

    return (SepConv, DilConv, FusedSequential)


_TAKES_PENDING = _takes_pending_types()  # ops whose first layer is a conv chain (it applies a pending input itself)


def _applies_pending(op):
    """Does ``op`` apply a pending BatchNorm + activation of its input as it loads - a conv at its head with no
    activation in front (SepConv, the dense conv + BN + ReLU ops)?  DilConv starts with a ReLU of its own and takes the
    finished map, like the pooling / skip / global-pool ops."""
    from .layer_factory import SepConv
    from .modules import _chainable, _flatten

    if isinstance(op, SepConv):
        return True
    if isinstance(op, FusedSequential):
        mods = _flatten(op._modules.values())
        return bool(mods) and _chainable(mods[0])
    return False


class _Handles(object):
    """The consumers' handles of a list of nodes (functional.fan_out): a node with several consumers is fanned out
    once, when its first consumer asks, and every consumer takes its own alias - in backward their gradients meet in
    ONE junction launch instead of autograd's pairwise adds.  ``finished[idx]``: per consumer of node idx, in
    consumption order, whether it needs the finished map of a node that is still pending."""

    def __init__(self, nodes, finished):
        self.nodes, self.finished, self.out = nodes, finished, {}

    def take(self, idx):
        flags = self.finished[idx]
        if len(flags) < 2:
            node = self.nodes[idx]
            return F.materialize(node) if (flags and flags[0]) else node
        queue = self.out.get(idx)
        if queue is None:
            node = self.nodes[idx]
            if not isinstance(node, F.Pending):
                flags = [False] * len(flags)
            n_fin = sum(flags)
            got = F.fan_out(node, len(flags) - n_fin, n_fin)
            raw, fin = got[:len(flags) - n_fin], got[len(flags) - n_fin:]
            queue = self.out[idx] = [fin.pop(0) if f else raw.pop(0) for f in flags]
        if not queue:
            # (more consumers than were counted: the extra one would read a node whose alias already went to
            #  somebody else, and its gradient would never reach the junction)
            raise F.NassegError("node {} has more consumers than the decoder counted ({})".format(idx, len(flags)))
        return queue.pop(0)

    def check_all_taken(self):
        """every handle that was made has a consumer: an alias nobody reads leaves its share out of the junction's
        sum without a word (a collect index listed twice, say).  Called at the end of a decoder's forward."""
        left = dict((idx, len(q)) for idx, q in self.out.items() if q)
        if left:
            raise F.NassegError("decoder nodes fanned out to consumers that never came: {}".format(left))


def _hw(t):
    return (int(t.size(2)), int(t.size(3)))


def collect_all(feats, collect_indices, relu=False):
    """Concatenate the selected maps along channels at the largest *height*
    among them (reference: micro_decoders.py:11-25).

    The reference walks the list and, whenever the running concat is lower than
    the next map, up-samples the *whole running concat*; a map that entered
    early may therefore be interpolated more than once.  That order is kept:
    each piece carries its own chain of target sizes, all hops but the last are
    materialised, and the last hop is fused with the write into the output slab
    (plus the ReLU the decoders apply right after, when ``relu`` is set).
    """
    first = feats[collect_indices[0]]
    cur = _hw(first)
    pieces = [[first, []]]
    for i in collect_indices[1:]:
        t = feats[i]
        size = _hw(t)
        if cur[0] > size[0]:
            pieces.append([t, [cur]])
        elif size[0] > cur[0]:
            for piece in pieces:
                piece[1].append(size)
            cur = size
            pieces.append([t, []])
        else:
            if size[1] != cur[1]:
                raise RuntimeError("Sizes of tensors must match except in dimension 1")
            pieces.append([t, []])
    if len(pieces) == 1 and not relu:
        return first
    ready = []
    for t, chain in pieces:
        for hop in chain[:-1]:
            t = F.bilinear_resize(t, hop)
        ready.append(t)
    return F.concat_resize(ready, cur, relu=relu)


def _sum_to_larger(x1, x2):
    """Bilinearly up-sample the (lexicographically) smaller map and add.  (Operands may be functional.Pending:
    F.add applies a pending BatchNorm + ReLU as it loads; a resize needs the finished map.)"""
    s1, s2 = _hw(x1), _hw(x2)
    if s1 > s2:
        x2 = F.bilinear_resize(F.materialize(x2), s1)
    elif s1 < s2:
        x1 = F.bilinear_resize(F.materialize(x1), s2)
    return F.add(x1, x2)


def _param_header(n_params):
    """first block of the genotype description both decoders print (millions of parameters)"""
    return "#PARAMS\n\n {:3.2f}M".format(n_params / 1e6)


def _attach_heads(decoder, collected_width, agg_size, num_classes):
    """The tail both decoders share: ``pre_clf`` (1x1 + BN + ReLU over the concatenated collected
    maps) and ``conv_clf`` (3x3 with bias, one channel per class); sets ``num_classes``."""
    decoder.pre_clf = conv_bn_relu(collected_width, agg_size, 1, 1, 0)
    decoder.conv_clf = conv3x3(agg_size, num_classes, stride=1, bias=True)
    decoder.num_classes = num_classes


class AggregateCell(nn.Module):
    """Optional 1x1 conv+BN+ReLU per branch, up-sample the smaller, add
    (micro_decoders.py:28-51)."""

    def __init__(self, size_1, size_2, agg_size, pre_transform=True):
        super(AggregateCell, self).__init__()
        self.pre_transform = pre_transform
        if pre_transform:  # one 1x1 projection per input, registered as branch_1 / branch_2
            for slot, width in enumerate((size_1, size_2), start=1):
                self.add_module("branch_{}".format(slot), conv_bn_relu(width, agg_size, 1, 1, 0))

    def forward(self, x1, x2):
        if self.pre_transform:
            x1, x2 = self.branch_1(x1), self.branch_2(x2)
        return _sum_to_larger(x1, x2)


class ContextualCell(nn.Module):
    """DAG of registry ops at constant resolution / width (micro_decoders.py:54-121).

    config = [op0, [pos1, pos2, op1, op2], ...]: op0 is applied to the input;
    every further entry applies two ops to earlier nodes and sums them.  Nodes
    nobody consumes ("loose ends") are summed into the output.
    """

    def __init__(self, config, inp, repeats=1):
        super(ContextualCell, self).__init__()
        self._ops = nn.ModuleList()
        self._pos = []
        self._collect_inds = [0]
        self._pools = ["x"]

        def add_op(op_id, src):
            name = OP_NAMES[op_id]
            if src in self._collect_inds:
                self._collect_inds.remove(src)
            self._ops.append(OPS[name](inp, inp, 1, True, repeats))
            self._pos.append(src)
            self._pools.append("{}({})".format(name, self._pools[src]))

        for step, entry in enumerate(config):
            if step == 0:
                add_op(entry, 0)
                self._collect_inds.append(1)
                continue
            src_a, src_b, op_a, op_b = entry
            add_op(op_a, src_a)
            add_op(op_b, src_b)
            node_a, node_b = step * 3 - 1, step * 3
            self._ops.append(AggregateCell(size_1=None, size_2=None, agg_size=inp,
                                           pre_transform=False))
            self._pos.append([node_a, node_b])
            self._collect_inds.append(step * 3 + 1)
            self._pools.append("sum({},{})".format(self._pools[node_a], self._pools[node_b]))

    def _consumers(self):
        """per node, in the order forward() consumes it: does that consumer need the FINISHED map of a pending node
        (an op that does not apply a pending input itself), or does it take the node as it is (ops with a conv at
        their head, the sums of a step, the sum of the loose ends)?"""
        table = getattr(self, "_consumer_table", None)
        if table is None:
            table = [[] for _ in range(len(self._ops) + 1)]
            for src, op in zip(self._pos, self._ops):
                if isinstance(src, list):
                    for node in src:
                        table[node].append(False)
                else:
                    table[src].append(not _applies_pending(op))
            for node in self._collect_inds:
                table[node].append(False)
            self._consumer_table = table
        return table

    def forward(self, x):
        # Ops that end in conv + BatchNorm + ReLU hand their raw conv output over with the tail pending
        # (functional.Pending): the sums of a step and of the loose ends apply it as they load, an op that starts
        # with a conv takes it as its prologue, anything else gets the normalised map (computed once per node).
        # A node with several consumers is fanned out (_Handles): their gradients meet in one junction launch,
        # which for a pending node also is the mask-and-reduce pass of its producer's BatchNorm backward.
        nodes = [x]
        handles = _Handles(nodes, self._consumers())
        for src, op in zip(self._pos, self._ops):
            if isinstance(src, list):
                assert len(src) == 2, "Two ops must be provided"
                nodes.append(op(handles.take(src[0]), handles.take(src[1])))
            else:
                inp = handles.take(src)
                if not isinstance(op, _TAKES_PENDING):
                    inp = F.materialize(inp)
                nodes.append(run_op(op, inp, defer_tail=True))
        out = None
        for i in self._collect_inds:
            node = handles.take(i)
            out = node if out is None else F.add(out, node)
        handles.check_all_taken()
        return F.materialize(out)

    def prettify(self):
        return " + ".join(self._pools[i] for i in self._collect_inds)


class MergeCell(nn.Module):
    """Two contextual cells (separate weights) joined by an AggregateCell
    (micro_decoders.py:124-139)."""

    def __init__(self, ctx_config, conn, inps, agg_size, ctx_cell, repeats=1):
        super(MergeCell, self).__init__()
        self.index_1, self.index_2 = conn
        # same genotype, separate weights, one cell per incoming map
        for slot, width in enumerate(inps, start=1):
            self.add_module("op_{}".format(slot), ctx_cell(ctx_config, width, repeats=repeats))
        self.agg = AggregateCell(inps[0], inps[1], agg_size)

    def forward(self, x1, x2):
        return self.agg(self.op_1(x1), self.op_2(x2))

    def prettify(self):
        return self.op_1.prettify()


class MicroDecoder(nn.Module):
    """CVPR'19 decoder (micro_decoders.py:142-254).

    ``forward(list_of_encoder_maps) -> (logits, [aux_logits per cell])``.
    NOTE (kept from the reference, :184): the constructor overwrites the entries
    of the caller's ``inp_sizes`` list with ``agg_size``.
    """

    def __init__(self, inp_sizes, num_classes, config, agg_size=64, num_pools=4,
                 ctx_cell=ContextualCell, aux_cell=False, repeats=1, **kwargs):
        super(MicroDecoder, self).__init__()
        self.aux_cell = aux_cell
        self.pool = ["l{}".format(i + 1) for i in range(num_pools)]
        self.agg_size = agg_size
        for n, width in enumerate(inp_sizes):
            setattr(self, "adapt{}".format(n + 1), conv_bn_relu(width, agg_size, 1, 1, 0, affine=True))
            inp_sizes[n] = agg_size
        widths = list(inp_sizes)
        cell_config, conns = config
        self.conns = conns
        self.ctx = cell_config
        self.repeats = repeats
        self.ctx_cell = ctx_cell
        self.collect_inds = []
        cells, heads = [], []
        for block, (ind_1, ind_2) in enumerate(conns):
            for ind in (ind_1, ind_2):
                if ind in self.collect_inds:
                    self.collect_inds.remove(ind)
            cells.append(MergeCell(cell_config, (ind_1, ind_2), (widths[ind_1], widths[ind_2]),
                                   agg_size, ctx_cell, repeats=repeats))
            head = FusedSequential()
            if aux_cell:
                head.add_module("aux_cell", ctx_cell(cell_config, agg_size, repeats=repeats))
            head.add_module("aux_clf", conv3x3(agg_size, num_classes, stride=1, bias=True))
            heads.append(head)
            self.collect_inds.append(block + num_pools)
            widths.append(agg_size)
            self.pool.append("({} + {})".format(self.pool[ind_1], self.pool[ind_2]))
        self.cells = nn.ModuleList(cells)
        self.aux_clfs = nn.ModuleList(heads)
        _attach_heads(self, agg_size * len(self.collect_inds), agg_size, num_classes)
        self.info = " + ".join(self.pool[i] for i in self.collect_inds)

    def prettify(self, n_params):
        return "\n\n".join((_param_header(n_params), "#Contextual:\n" + self.cells[0].prettify(),
                            "#Connections:\n" + self.info))

    def forward(self, x):
        maps = [getattr(self, "adapt{}".format(n + 1))(t) for n, t in enumerate(x)]
        # consumers of every map, in order: the cells' inputs, the auxiliary head of a cell's output, collect_all
        uses = [[] for _ in range(len(maps) + len(self.conns))]
        for block, (a, b) in enumerate(self.conns):
            uses[a].append(False)
            uses[b].append(False)
            uses[len(x) + block].append(False)  # (its auxiliary classifier)
        for i in self.collect_inds:
            uses[i].append(False)
        handles = _Handles(maps, uses)
        aux_outs = []
        for block, (cell, head, (a, b)) in enumerate(zip(self.cells, self.aux_clfs, self.conns)):
            maps.append(cell(handles.take(a), handles.take(b)))
            aux_outs.append(head(handles.take(len(x) + block)))
        # F.relu(collect_all(...)) of the reference: the ReLU is applied as pre_clf loads
        picked = {i: handles.take(i) for i in self.collect_inds}
        handles.check_all_taken()
        out = collect_all(picked, self.collect_inds)
        return self.conv_clf(self.pre_clf(out, relu_in=True)), aux_outs


class TemplateDecoder(nn.Module):
    """WACV'20 decoder with template repeats and strides (micro_decoders.py:257-398).

    ``forward(list_of_encoder_maps) -> logits``.  Blocks in the first half of
    ``structure`` aggregate at the smaller of their two resolutions, the rest at
    the larger; a block's width is its input width times ``stride**stride_power``.
    """

    def __init__(self, inp_sizes, num_classes, config, agg_size=64, num_pools=4, repeats=1,
                 stride_power=1, **kwargs):
        super(TemplateDecoder, self).__init__()
        widths = list(inp_sizes)
        n_scales = len(widths)
        templates, structure = config
        n_blocks = len(structure)
        widths += [0] * n_blocks
        self.agg_size = agg_size
        self._ops = nn.ModuleList()
        self._pos = []
        self._collect_inds = []
        self._repeats = []
        self._pools = ["l{}".format(j + 1) for j in range(n_scales)]

        for block, (pos1, pos2, cell_id, n_rep, stride_log2) in enumerate(structure):
            larger = block >= (n_blocks // 2)
            n_rep += 1  # genotype stores repeats zero-based
            stride = 2 ** stride_log2
            op_id1, op_id2, agg_id = templates[cell_id]
            block_ops = nn.ModuleList()
            block_pos = []
            agg_width = None
            out_w = [0, 0]
            in_w = [0, 0]
            for rep in range(n_rep):
                for branch, (pos, op_id) in enumerate(((pos1, op_id1), (pos2, op_id2))):
                    if rep == 0:
                        cin = widths[pos]
                        cout = cin * int(stride ** stride_power)
                    elif branch == 0:
                        cin = cout = in_w[-1]
                    else:
                        cin = cout = agg_width
                    out_w[branch] = cout
                    in_w[branch] = cin
                    if pos in self._collect_inds:
                        self._collect_inds.remove(pos)
                    name = OP_NAMES_WACV[op_id]
                    block_ops.append(OPS[name](cin, cout, stride, True, repeats=repeats))
                    block_pos.append(pos)
                    self._pools.append("{}({})".format(name, self._pools[pos]))
                agg_name = AGG_OP_NAMES[agg_id]
                agg_width = max(out_w)
                block_ops.append(AGG_OPS[agg_name](out_w[0], out_w[1], agg_width, True,
                                                   repeats=repeats, larger=larger))
            node = n_scales + block
            widths[node] = agg_width
            self._pos.append(block_pos)
            self._ops.append(block_ops)
            self._repeats.append(n_rep)
            self._collect_inds.append(node)
            self._pools.append("{}({},{})".format(agg_name, self._pools[n_scales + block - 2],
                                                  self._pools[n_scales + block - 1]))
        _attach_heads(self, sum(widths[idx] for idx in self._collect_inds), agg_size, num_classes)
        self.info = " + ".join(self._pools[i] for i in self._collect_inds)

    def _reset_clf(self, num_classes):
        """Swap the classifier for a different label set (micro_decoders.py:367-373).
        The reference reads an attribute it never sets and hard-codes ``.cuda()``;
        here the new head is created on the device of the old one."""
        if num_classes != self.num_classes:
            device = self.conv_clf.weight.device
            del self.conv_clf
            self.conv_clf = conv3x3(self.agg_size, num_classes, stride=1, bias=True).to(device)
            self.num_classes = num_classes
            TREE_VERSION[0] += 1  # (cached parameter lists of the engine are stale now)

    def prettify(self, n_params):
        return _param_header(n_params) + "\n\n#Connections:\n" + self.info

    def forward(self, x):
        maps = list(x)
        # Every value with several consumers - an encoder map or a block's output read by several blocks / repeats and
        # by collect_all, a repeat's output read by the next two repeats - is fanned out (functional.fan_out): the
        # consumers' gradients meet in one junction launch instead of autograd's pairwise adds.  Values are numbered
        # as they appear: the maps first, then every repeat's output (a block's last one doubles as its map).
        n_maps = len(maps) + len(self._pos)
        uses = [[] for _ in range(n_maps)]
        plan = []  # per block: [(left value, right value, output value) per repeat]
        for block, (pos, n_rep) in enumerate(zip(self._pos, self._repeats)):
            assert isinstance(pos, list), "Must be list"
            left, right = pos[0], pos[1]
            reps = []
            for rep in range(n_rep):
                if rep == n_rep - 1:
                    out_id = len(x) + block
                else:
                    out_id = len(uses)
                    uses.append([])
                uses[left].append(False)
                uses[right].append(False)
                reps.append((left, right, out_id))
                left, right = right, out_id
            plan.append(reps)
        for i in self._collect_inds:
            uses[i].append(False)
        values = maps + [None] * (len(uses) - len(maps))
        handles = _Handles(values, uses)
        for reps, ops in zip(plan, self._ops):
            for rep, (left, right, out_id) in enumerate(reps):
                # (an aggregation op that applies its producers' last BatchNorm + ReLU as it loads gets their
                #  raw conv outputs: the normalised maps are never written)
                defer = getattr(ops[rep * 3 + 2], "accepts_pending", False)
                a = run_op(ops[rep * 3], handles.take(left), defer)
                b = run_op(ops[rep * 3 + 1], handles.take(right), defer)
                # the next repeat consumes (previous right input, previous output)
                values[out_id] = ops[rep * 3 + 2](a, b)
        # F.relu(collect_all(...)) of the reference: the ReLU is applied as pre_clf loads
        picked = {i: handles.take(i) for i in self._collect_inds}
        handles.check_all_taken()
        out = collect_all(picked, self._collect_inds)
        return self.conv_clf(self.pre_clf(out, relu_in=True))
