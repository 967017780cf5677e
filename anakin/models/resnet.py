from artiboost_amd.hpregnet import ResNet18, ResNet34  # noqa: F401  (anakin/models/resnet.py:240,249: the torch backbones of the regression model)
