ConvertType = dict
