from . import channel, instance  # noqa: F401
