from .audio_extractor import Extractor as Extractor
from .audio_feature_extractor import ConvFeatureExtractor as ConvFeatureExtractor
