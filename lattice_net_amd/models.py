"""`LNN`: the LatticeNet U-Net over lattice levels (reference latticenet_py/lattice/models.py:70-266), assembled from
the operator modules of this package — distribute -> PointNet -> {ResNet / bottleneck blocks, coarsen} x downsamples ->
bottleneck blocks -> {finefy, skip concat, blocks} x downsamples -> DeformSlice classification head.

Attribute names of sub-modules match the reference (`resnet_blocks_per_down_lvl_list`, `coarsens_list`,
`finefy_list`, `slice_fast_cuda`, ...) so `state_dict` keys carry over.  `prepare_cloud` (models.py:18-66) is the
cloud -> tensors glue; it accepts any object with numpy attributes V / C / I / L_gt.
"""
from __future__ import annotations

import numpy as np
import torch

from .lattice_blocks import BottleneckBlock, CoarsenAct, GnReluFinefy, ResnetBlock, SliceFastCUDALatticeModule
from .lattice_modules import DistributeLatticeModule, PointNetModule

__all__ = ["LNN", "prepare_cloud"]


def prepare_cloud(cloud, model_params, device="cuda"):
    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a)).float().to(device)

    with torch.no_grad():
        pmode = model_params.positions_mode()
        if pmode == "xyz":
            positions = t(cloud.V)
        elif pmode == "xyz+rgb":
            positions = torch.cat((t(cloud.V), t(cloud.C)), 1)
        elif pmode == "xyz+intensity":
            positions = torch.cat((t(cloud.V), t(cloud.I)), 1)
        else:
            raise ValueError(f"positions mode {pmode!r} not implemented")
        vmode = model_params.values_mode()
        if vmode == "none":
            values = torch.zeros((positions.shape[0], 1), device=device)  # an (ignored) single channel, models.py:38-39
        elif vmode == "intensity":
            values = t(cloud.I)
        elif vmode == "rgb":
            values = t(cloud.C)
        elif vmode == "rgb+height":
            values = torch.cat((t(cloud.C), t(cloud.V[:, 1:2])), 1)
        elif vmode == "rgb+xyz":
            values = torch.cat((t(cloud.C), t(cloud.V)), 1)
        elif vmode == "height":
            values = t(cloud.V[:, 1:2])
        elif vmode == "xyz":
            values = t(cloud.V)
        else:
            raise ValueError(f"values mode {vmode!r} not implemented")
        target = torch.from_numpy(np.ascontiguousarray(cloud.L_gt)).long().reshape(-1).to(device)
    return positions.contiguous(), values.contiguous(), target


class LNN(torch.nn.Module):
    def __init__(self, nr_classes, model_params, device="cuda"):
        super().__init__()
        self.nr_classes = nr_classes
        self.model_params = model_params
        self.nr_downsamples = model_params.nr_downsamples()
        self.nr_blocks_down_stage = model_params.nr_blocks_down_stage()
        self.nr_blocks_bottleneck = model_params.nr_blocks_bottleneck()
        self.nr_blocks_up_stage = model_params.nr_blocks_up_stage()
        self.nr_levels_down_with_normal_resnet = model_params.nr_levels_down_with_normal_resnet()
        self.nr_levels_up_with_normal_resnet = model_params.nr_levels_up_with_normal_resnet()
        compression_factor = model_params.compression_factor()
        experiment = "none"  # models.py:102 pins it regardless of the cfg

        self.distribute = DistributeLatticeModule()
        self.pointnet_channels_per_layer = model_params.pointnet_channels_per_layer()
        self.start_nr_filters = model_params.pointnet_start_nr_channels()
        # distributed rows are [positions | values | barycentric weight]; the MLP sees all but the last column.  The
        # reference creates these layers inside the first forward (mods:636-651); the width only depends on the cfg,
        # so they exist from construction here (checkpoints load, optimizers see them, before any forward)
        pos_ch = {"xyz": 3, "xyz+rgb": 6, "xyz+intensity": 4}.get(model_params.positions_mode())
        val_ch = {"none": 1, "intensity": 1, "rgb": 3, "rgb+height": 4, "rgb+xyz": 6, "height": 1, "xyz": 3}.get(model_params.values_mode())
        nr_in = pos_ch + val_ch if (pos_ch is not None and val_ch is not None) else None
        self.point_net = PointNetModule(self.pointnet_channels_per_layer, self.start_nr_filters, nr_input_channels=nr_in, device=device)

        # ---- encoder
        self.resnet_blocks_per_down_lvl_list = torch.nn.ModuleList([])
        self.coarsens_list = torch.nn.ModuleList([])
        skip_channels = []
        cur = self.start_nr_filters
        for i in range(self.nr_downsamples):
            stage = torch.nn.ModuleList([])
            for _ in range(self.nr_blocks_down_stage[i]):
                if i < self.nr_levels_down_with_normal_resnet:
                    stage.append(ResnetBlock(cur, cur, [1, 1], [False, False], False, device=device))
                else:
                    stage.append(BottleneckBlock(cur, cur, [False, False, False], device=device))
            self.resnet_blocks_per_down_lvl_list.append(stage)
            skip_channels.append(cur)
            after = int(cur * 2 * compression_factor)
            self.coarsens_list.append(CoarsenAct(cur, after, device=device))
            cur = after

        # ---- bottleneck
        self.resnet_blocks_bottleneck = torch.nn.ModuleList(
            [BottleneckBlock(cur, cur, [False, False, False], device=device) for _ in range(self.nr_blocks_bottleneck)])

        # ---- decoder
        self.do_concat_for_vertical_connection = True
        self.finefy_list = torch.nn.ModuleList([])
        self.resnet_blocks_per_up_lvl_list = torch.nn.ModuleList([])
        for i in range(self.nr_downsamples):
            skip = skip_channels.pop()
            nr_finefy = int(cur / 2)
            self.finefy_list.append(GnReluFinefy(cur, nr_finefy, device=device))
            cur = skip + nr_finefy if self.do_concat_for_vertical_connection else skip
            stage = torch.nn.ModuleList([])
            for j in range(self.nr_blocks_up_stage[i]):
                # the very last convolution feeds the slice, not a norm: it carries a bias (models.py:176)
                is_last_conv = (j == self.nr_blocks_up_stage[i] - 1) and (i == self.nr_downsamples - 1)
                if i >= self.nr_downsamples - self.nr_levels_up_with_normal_resnet:
                    stage.append(ResnetBlock(cur, cur, [1, 1], [False, is_last_conv], False, device=device))
                else:
                    stage.append(BottleneckBlock(cur, cur, [False, False, is_last_conv], device=device))
            self.resnet_blocks_per_up_lvl_list.append(stage)

        self.slice_fast_cuda = SliceFastCUDALatticeModule(in_channels=cur, nr_classes=nr_classes,
                                                          dropout_prob=model_params.dropout_last_layer(), experiment=experiment,
                                                          device=device)
        self.logsoftmax = torch.nn.LogSoftmax(dim=1)

    def forward(self, ls, positions, values):
        with torch.no_grad():
            ls, distributed, indices, weights = self.distribute(ls, positions, values)
        lv, ls = self.point_net(ls, distributed, indices)

        fine_structures, fine_values = [], []
        for i in range(self.nr_downsamples):
            for block in self.resnet_blocks_per_down_lvl_list[i]:
                lv, ls = block(lv, ls)
            fine_structures.append(ls)
            fine_values.append(lv)
            lv, ls = self.coarsens_list[i](lv, ls)

        for block in self.resnet_blocks_bottleneck:
            lv, ls = block(lv, ls)

        for i in range(self.nr_downsamples):
            skip_values = fine_values.pop()
            fine_structure = fine_structures.pop()
            lv, ls = self.finefy_list[i](lv, ls, fine_structure)
            lv = torch.cat((lv, skip_values), 1) if self.do_concat_for_vertical_connection else lv + skip_values
            for block in self.resnet_blocks_per_up_lvl_list[i]:
                lv, ls = block(lv, ls)

        logits = self.slice_fast_cuda(lv, ls, positions, indices, weights)
        return self.logsoftmax(logits), logits
