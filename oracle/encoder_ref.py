"""ORACLE (test infrastructure) -- PyTorch-CPU restatement of the reference's image encoder and heads:
``InterHandEncoder`` (``src/models/networks.py:45-80``) = ResNet-50 trunk (``src/models/resnet.py:97-156``:
7x7/2 stem, 3x3/2 max-pool, bottleneck stages [3,4,6,3] with the stride on the 3x3 conv (``:65``),
``AvgPool2d(7)``, ReLU, ``fc1`` 2048->1024, ReLU) -> ``feat_encoder`` (ReLU, Linear 1024->1024, ReLU) ->
3 IEF iterations ``params += Linear(1146->122)([feat | params])`` from ``mean_params`` -> sigmoid(Linear 1024->2);
and ``InterHandSubNetwork`` (``networks.py:83-105``): 1146 -> 512 -> 256 -> 128 -> k with ReLU.

Parameter names equal the reference's ``state_dict`` keys, so a reference checkpoint loads unchanged.
Pinned by ``tests/golden/encoder.npz`` and ``mlp_head.npz`` (reference run on seeded weights/images).
"""
import torch
import torch.nn as nn


class _Bottleneck(nn.Module):
    def __init__(self, cin, planes, stride, project):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        y = torch.relu(self.bn1(self.conv1(x)))
        y = torch.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        r = x if self.downsample is None else self.downsample(x)
        return torch.relu(y + r)


class ResNet50Ref(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for li, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], start=1):
            layers = []
            for b in range(blocks):
                layers.append(_Bottleneck(cin, planes, stride if b == 0 else 1, project=(b == 0)))
                cin = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*layers))
        self.fc1 = nn.Linear(2048, 1024)

    def forward(self, x):
        x = torch.relu(self.bn1(self.conv1(x)))
        x = nn.functional.max_pool2d(x, 3, stride=2, padding=1)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = nn.functional.avg_pool2d(x, 7, stride=1).flatten(1)
        return torch.relu(self.fc1(torch.relu(x)))


class InterHandEncoderRef(nn.Module):
    def __init__(self, mean_params, total_params_dim=122):
        super().__init__()
        self.mean_params = mean_params.clone()
        self.main_encoder = ResNet50Ref()
        self.feat_encoder = nn.Sequential(nn.ReLU(), nn.Linear(1024, 1024), nn.ReLU())
        self.regressor_ih = nn.Sequential(nn.Linear(1024 + total_params_dim, total_params_dim))
        self.hand_classifier = nn.Sequential(nn.Linear(1024, 2))

    def forward(self, img):
        feat = self.feat_encoder(self.main_encoder(img))
        params = self.mean_params
        for _ in range(3):
            params = params + self.regressor_ih(torch.cat([feat, params], dim=1))
        return params, torch.sigmoid(self.hand_classifier(feat))


class InterHandSubNetworkRef(nn.Module):
    def __init__(self, input_dim, update_param_dim):
        super().__init__()
        self.regressor = nn.Sequential(nn.Linear(input_dim, 512), nn.ReLU(), nn.Linear(512, 256), nn.ReLU(),
                                       nn.Linear(256, 128), nn.ReLU(), nn.Linear(128, update_param_dim))

    def forward(self, x):
        return self.regressor(x)
