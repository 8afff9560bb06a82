from .NoiseIterator import NoiseIterator as NoiseIterator
from .SceneIterator import SceneIterator as SceneIterator
