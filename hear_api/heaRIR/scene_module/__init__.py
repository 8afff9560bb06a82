from .generate_scenes import generate_scene as generate_scene
