"""Import-path shim: ``yolo3.*`` names of the reference resolve to the MI355X-native package."""
