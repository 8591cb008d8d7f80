from .amp_conv import AMPConv, InvalidConfiguration, MessagePassing

__all__ = ['AMPConv', 'InvalidConfiguration', 'MessagePassing']
