"""TEST INFRASTRUCTURE ONLY.  Restatement of torchvision's public ResNet v1.5 (bottleneck family) for the convolutional student of
DistillationV3 (BASELINE.json configs[3]; reference wrapper LT/_models/torchvision/resnet.py:21-47).

torchvision is a third-party dependency of the reference (`pyproject.toml`: `torchvision>=0.15`), neither vendored under
/root/reference nor installed in this image, so its code cannot be compiled or imported here: **parity unpinned** for the
architecture itself.  What is restated (torchvision/models/resnet.py, `ResNet` / `Bottleneck`, v1.5 = stride on the 3x3
convolution): conv1 7x7/2 (no bias) - bn1 - relu - maxpool 3x3/2 pad 1 - layer1..4 of bottlenecks [1x1, 3x3(stride), 1x1 x4] with
a 1x1(stride) + BN downsample on the first block of a layer - avgpool - fc; BatchNorm2d eps 1e-5 / momentum 0.1;
kaiming_normal_(fan_out, relu) convolutions, BN weight 1 / bias 0; attribute names, hence state_dict keys, as torchvision's.
Anchors that ARE checked (tests/test_oracle_pin.py): the parameter count of resnet50 (25 557 032, torchvision's documented
figure) and the state_dict key list; and the reference's own `ResNetModelWrapper` + `DistillationV3` run on this class through
oracle/ref_harness.py (which registers it as `torchvision.models.ResNet`) to write tests/golden/distill_v3_resnet.pt."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Tuple

import torch
from torch import Tensor, nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: nn.Module | None = None) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x: Tensor) -> Tensor:
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet(nn.Module):
    def __init__(self, layers: Tuple[int, ...] = (3, 4, 6, 3), num_classes: int = 1000, width: int = 64) -> None:
        super().__init__()
        self.inplanes = width
        self.conv1 = nn.Conv2d(3, width, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(width, layers[0], 1)
        self.layer2 = self._make_layer(width * 2, layers[1], 2)
        self.layer3 = self._make_layer(width * 4, layers[2], 2)
        self.layer4 = self._make_layer(width * 8, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(width * 8 * Bottleneck.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes: int, blocks: int, stride: int) -> nn.Sequential:
        downsample = None
        if stride != 1 or self.inplanes != planes * Bottleneck.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * Bottleneck.expansion, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * Bottleneck.expansion))
        layers: List[nn.Module] = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * Bottleneck.expansion
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x: Tensor) -> Tensor:
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet50(**kw) -> ResNet:
    return ResNet((3, 4, 6, 3), **kw)


class IntermediateLayerGetter(nn.ModuleDict):
    """torchvision.models._utils.IntermediateLayerGetter: the model's children in order up to the last requested one; forward
    returns {new_name: output} for the requested layers."""

    def __init__(self, model: nn.Module, return_layers: Dict[str, str]) -> None:
        remaining = dict(return_layers)
        layers: "OrderedDict[str, nn.Module]" = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = dict(return_layers)

    def forward(self, x: Tensor) -> Dict[str, Tensor]:
        out: "OrderedDict[str, Tensor]" = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


def features(model: ResNet, x: Tensor) -> Tensor:
    """`ResNetModelWrapper.forward_features(x)["features"]`: everything up to and including layer4, [B, C, h, w]."""
    x = model.maxpool(model.relu(model.bn1(model.conv1(x))))
    return model.layer4(model.layer3(model.layer2(model.layer1(x))))
