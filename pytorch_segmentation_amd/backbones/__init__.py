from .mobilenet import MobileNetV2, mobilenet_v2
from .resnet import ResNet50, resnet50

__all__ = ['ResNet50', 'resnet50', 'MobileNetV2', 'mobilenet_v2']
