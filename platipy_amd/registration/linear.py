"""placeholder replaced below"""


def linear_registration(*a, **k):
    raise NotImplementedError
