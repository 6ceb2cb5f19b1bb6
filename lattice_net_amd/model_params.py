"""`ModelParams`: the `model: { ... }` block of a LatticeNet cfg file (reference src/ModelParams.cxx:22-53, bound to
Python at src/PyBridge.cxx:139-152) with the same accessor methods.

The reference reads `pointnet_channels_per_layer` (ModelParams.cxx:41) while its own shipped configs spell the key
`pointnet_layers` (config/lnn_train_semantic_kitti.cfg:38): both are accepted here.
"""
from __future__ import annotations

import os
import re
from typing import List

__all__ = ["ModelParams", "read_cfg_block"]


def _strip_comments(text: str) -> str:
    out = []
    for line in text.splitlines():
        in_str = False
        cut = len(line)
        for i, ch in enumerate(line):
            if ch == '"':
                in_str = not in_str
            elif not in_str and line.startswith("//", i):
                cut = i
                break
        out.append(line[:cut])
    return "\n".join(out)


def _convert(token: str):
    token = token.strip()
    if token.startswith('"') and token.endswith('"'):
        return token[1:-1]
    if token.startswith("["):
        inner = token[1:-1].strip()
        return [_convert(t) for t in inner.split(",") if t.strip()] if inner else []
    if token in ("true", "false"):
        return token == "true"
    try:
        return int(token)
    except ValueError:
        try:
            return float(token)
        except ValueError:
            return token


def read_cfg_block(path: str, block: str) -> dict:
    """Key/value pairs of one top-level `block: { ... }` of a configuru CFG file: numbers, quoted strings, booleans and
    flat lists; `//` comments ignored; nested blocks are skipped."""
    with open(path, "r") as f:
        text = _strip_comments(f.read())
    m = re.search(r"(?<![\w])" + re.escape(block) + r"\s*:\s*\{", text)
    if not m:
        raise ValueError(f"{path}: no `{block}` block")
    depth, i = 1, m.end()
    start = i
    while i < len(text) and depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    if depth:
        raise ValueError(f"{path}: unterminated `{block}` block")
    body = text[start:i - 1]
    body = re.sub(r"\w+\s*:\s*\{[^{}]*\}", "", body)  # nested blocks
    out = {}
    for key, val in re.findall(r"(\w+)\s*:\s*(\"[^\"]*\"|\[[^\]]*\]|[^\s,\[\]{}]+)", body):
        out[key] = _convert(val)
    return out


class ModelParams:
    def __init__(self, config_file: str):
        path = config_file
        if not os.path.isabs(path) and not os.path.exists(path):
            path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), config_file)
        cfg = read_cfg_block(path, "model")

        def need(*names):
            for nm in names:
                if nm in cfg:
                    return cfg[nm]
            raise KeyError(f"{path}: model block has no `{names[0]}`")

        self.m_positions_mode = str(need("positions_mode"))
        self.m_values_mode = str(need("values_mode"))
        self.m_pointnet_channels_per_layer = [int(x) for x in need("pointnet_channels_per_layer", "pointnet_layers")]
        self.m_pointnet_start_nr_channels = int(need("pointnet_start_nr_channels"))
        self.m_nr_downsamples = int(need("nr_downsamples"))
        self.m_nr_blocks_down_stage = [int(x) for x in need("nr_blocks_down_stage")]
        self.m_nr_blocks_bottleneck = int(need("nr_blocks_bottleneck"))
        self.m_nr_blocks_up_stage = [int(x) for x in need("nr_blocks_up_stage")]
        self.m_nr_levels_down_with_normal_resnet = int(need("nr_levels_down_with_normal_resnet"))
        self.m_nr_levels_up_with_normal_resnet = int(need("nr_levels_up_with_normal_resnet"))
        self.m_compression_factor = float(need("compression_factor"))
        self.m_dropout_last_layer = float(need("dropout_last_layer"))
        self.m_experiment = str(cfg.get("experiment", "none"))

    @staticmethod
    def create(config_file: str) -> "ModelParams":  # PyBridge.cxx:140
        return ModelParams(config_file)

    def positions_mode(self) -> str:
        return self.m_positions_mode

    def values_mode(self) -> str:
        return self.m_values_mode

    def pointnet_channels_per_layer(self) -> List[int]:
        return list(self.m_pointnet_channels_per_layer)

    def pointnet_start_nr_channels(self) -> int:
        return self.m_pointnet_start_nr_channels

    def nr_downsamples(self) -> int:
        return self.m_nr_downsamples

    def nr_blocks_down_stage(self) -> List[int]:
        return list(self.m_nr_blocks_down_stage)

    def nr_blocks_bottleneck(self) -> int:
        return self.m_nr_blocks_bottleneck

    def nr_blocks_up_stage(self) -> List[int]:
        return list(self.m_nr_blocks_up_stage)

    def nr_levels_down_with_normal_resnet(self) -> int:
        return self.m_nr_levels_down_with_normal_resnet

    def nr_levels_up_with_normal_resnet(self) -> int:
        return self.m_nr_levels_up_with_normal_resnet

    def compression_factor(self) -> float:
        return self.m_compression_factor

    def dropout_last_layer(self) -> float:
        return self.m_dropout_last_layer

    def experiment(self) -> str:
        return self.m_experiment
