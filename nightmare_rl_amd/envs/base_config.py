"""BaseConfig: nested config classes become mutable instances (mirrors reference envs/base_config.py:3-25)."""
import inspect


class BaseConfig:
    def __init__(self) -> None:
        self.init_member_classes(self)

    @staticmethod
    def init_member_classes(obj):
        for key in dir(obj):
            if key == "__class__":
                continue
            var = getattr(obj, key)
            if inspect.isclass(var):
                inst = var()
                setattr(obj, key, inst)
                BaseConfig.init_member_classes(inst)
